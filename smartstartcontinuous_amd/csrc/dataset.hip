// dataset.hip -- random-rollout data collection -> dynamics-model training set, entirely in HBM.
//
// The reference collects `num_rollouts` random-policy rollouts of `steps_per_rollout` steps, each stopping at
// the first terminal step (NN_Dynamics_Model/collect_samples_threaded.py:52-111), then turns the list of
// rollouts into (dataX, dataY, dataZ) = (s_i, a_i, s_{i+1} - s_i) with the last entry of every rollout dropped
// (NN_Dynamics_Model/data_manipulation.py:58-88), z-scores the three arrays (NND_MB_agent.py:302-319) and may
// add signal-proportional Gaussian noise (helper_funcs.py:10-17).  Here a rollout is one env's FIRST episode
// segment of a transition chunk written by ssc_rollout (one env per rollout, all of them in one launch):
//
//   ssc_dataset_scan    len[i]  = steps of rollout i (first done inclusive, else K);  off = exclusive prefix sum
//                       of (len - 1) = first row of rollout i in the row-major data set (off[n] = total rows).
//   ssc_dataset_build   compacts the SoA chunk columns [K][n] into row-major dataX [rows][obs_dim],
//                       dataY [rows][1], dataZ [rows][obs_dim], rollout-major like np.concatenate over the list.
//                       64 x 64 (step x env) tiles through LDS: reads coalesced along envs, writes along rows.
//   ssc_column_stats    per-column mean and population std of a row-major fp32 matrix, accumulated in f64 with
//                       a fixed summation order (partials per block, summed in block order) => bit-reproducible.
//   ssc_zscore          out[:, col0 + c] = nan_to_num((x[:, c] - mean[c]) / std[c]) evaluated in f64.
//   ssc_add_noise       x[:, c] += |mean[c] * noise_to_signal| * N(0,1) where mean[c] * noise_to_signal > 0.
//
// All of it is HBM-bound byte moving: (2*obs_dim + 1) * 4 B read (+1 B done) and written per data-set row.
#include <type_traits>

#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

enum : uint32_t { TAG_DATA_NOISE = 6 };

// ---------------------------------------------------------------------------------------------------------
// Rollout lengths.  One block = 64 envs (lane = env, coalesced 64-byte rows of the done column) x 4 waves, each
// wave scanning a quarter of the K steps 8 loads at a time (the loop is latency-bound, not bandwidth-bound:
// K dependent round trips otherwise); the quarters meet in an LDS atomicMin.  Also leaves the block's row count.
constexpr int kGroup = 64;     // envs per block in the scan / build kernels
__global__ __launch_bounds__(256) void dataset_len_kernel(const uint8_t *__restrict__ done, int64_t drs, int32_t K,
                                                           int64_t n, int32_t *__restrict__ len,
                                                           int64_t *__restrict__ group_rows) {
    __shared__ int32_t first[kGroup];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kGroup + lane;
    const bool live = i < n;
    if (w == 0) first[lane] = K;
    __syncthreads();
    const int32_t per = (K + 3) / 4;
    const int32_t kb = w * per, ke = kb + per < K ? kb + per : K;
    int32_t L = K;
    for (int32_t k0 = kb; k0 < ke && L == K; k0 += 8) {
        uint8_t d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = (live && k0 + j < ke) ? done[(int64_t)(k0 + j) * drs + i] : (uint8_t)0;
#pragma unroll
        for (int j = 7; j >= 0; --j)
            if (d[j] != 0) L = k0 + j + 1;
        if (__ballot(L == K && live) == 0) break;      // every lane of the wave has its first terminal step
    }
    if (L < K) atomicMin(&first[lane], L);
    __syncthreads();
    if (w == 0) {
        const int32_t Lf = first[lane];
        if (live) len[i] = Lf;
        int64_t rows = live && Lf > 1 ? Lf - 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rows += __shfl_xor(rows, o);
        if (lane == 0) group_rows[blockIdx.x] = rows;
    }
}

// One block: exclusive prefix sum of the per-group row counts, in place (group_rows[g] becomes the first row of
// group g); total -> *total_out.
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void dataset_group_scan_kernel(int64_t *__restrict__ group_rows, int64_t ngroups,
                                                                           int64_t *__restrict__ total_out) {
    __shared__ int64_t part[kScanThreads];
    const int t = threadIdx.x;
    const int64_t span = (ngroups + kScanThreads - 1) / kScanThreads;
    const int64_t b = (int64_t)t * span, e = b + span < ngroups ? b + span : ngroups;
    int64_t s = 0;
    for (int64_t g = b; g < e; ++g) s += group_rows[g];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < kScanThreads; d <<= 1) {      // Hillis-Steele inclusive scan over the 1024 partial sums
        const int64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = t > 0 ? part[t - 1] : 0;
    for (int64_t g = b; g < e; ++g) {
        const int64_t r = group_rows[g];
        group_rows[g] = run;
        run += r;
    }
    if (t == kScanThreads - 1) *total_out = part[kScanThreads - 1];
}

// off[i] = group base + exclusive scan of max(len - 1, 0) inside the group of 64 envs (one wave per group).
__device__ __forceinline__ int64_t wave_exclusive_rows(int32_t rows, int lane) {
    int64_t incl = rows;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    return incl - rows;
}

__global__ __launch_bounds__(64) void dataset_offsets_kernel(const int32_t *__restrict__ len, const int64_t *__restrict__ group_base,
                                                              int64_t n, int64_t *__restrict__ off) {
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kGroup + lane;
    const int32_t rows = i < n && len[i] > 1 ? len[i] - 1 : 0;
    const int64_t o = group_base[blockIdx.x] + wave_exclusive_rows(rows, lane);
    if (i < n) off[i] = o;
}

struct DatasetArgs {
    ssc_transition_log log;
    const int32_t *len;
    const int64_t *off;
    float *X, *Y, *Z;
    int64_t n, rs, capacity;
};

// grid (ceil(n / 64), ceil((K - 1) / 64)); 256 threads = 4 waves; one block = 64 envs x 64 steps.
// Reads are coalesced along envs (lane = env: 256 contiguous bytes of one step of one column); the D columns of a
// matrix are staged in LDS and written out with lane = flattened (step, column) index of ONE env, i.e. 256
// contiguous bytes of the row-major output.  The observation values stay in registers between the dataX and the
// dataZ phase, so every chunk byte is read once.
constexpr int kTileK = 64;
constexpr int kTileStride = kTileK * 65;       // [step][env] with a one-word skew: transposed reads stay conflict-free
template <int D>
__global__ __launch_bounds__(256) void dataset_build_kernel(DatasetArgs g) {
    constexpr int kPad = (64 + D - 1) / D;     // column planes start kPad banks apart (lanes of a row differ in c)
    __shared__ float tile[D * (kTileStride + kPad)];
    __shared__ int32_t rows_s[kGroup];
    __shared__ int64_t off_s[kGroup];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t e0 = (int64_t)blockIdx.x * kGroup;
    const int32_t k0 = blockIdx.y * kTileK;
    int32_t rows = 0;
    if (w == 0) {
        const int64_t env = e0 + lane;
        rows = env < g.n ? (g.len[env] > 1 ? g.len[env] - 1 : 0) : 0;
        rows_s[lane] = rows;
        off_s[lane] = env < g.n ? g.off[env] : 0;
    }
    if (__syncthreads_or(rows > k0) == 0) return;     // no rollout of this env group reaches step k0
    const int64_t env = e0 + lane;
    const int32_t my_rows = rows_s[lane];
    // every chunk byte this block needs is requested up front (7 x 16 independent loads per thread at D = 3)
    float o[D][16], o2[D][16], a[16];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float *src = g.log.obs[c], *src2 = g.log.obs2[c];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int32_t k = k0 + w + 4 * j;
            o[c][j] = k < my_rows ? src[(int64_t)k * g.rs + env] : 0.0f;
            o2[c][j] = k < my_rows ? src2[(int64_t)k * g.rs + env] : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int32_t k = k0 + w + 4 * j;
        a[j] = k < my_rows ? g.log.act[(int64_t)k * g.rs + env] : 0.0f;
    }
    auto stage = [&](int c, int j, float v) { tile[c * (kTileStride + kPad) + (w + 4 * j) * 65 + lane] = v; };
    // one env per (wave, j): width*64 consecutive floats of the output starting at row off + k0
    // (the output width is a compile-time constant of each call: as a run-time argument every element paid an integer
    // division for its (row, column))
    auto flush = [&](float *__restrict__ dst, auto width_tag) {
        constexpr int width = decltype(width_tag)::value;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int el = w + 4 * j;
            const int32_t r = rows_s[el] - k0;            // rows of this env inside the tile (may be <= 0 or > 64)
            const int64_t base = (off_s[el] + k0) * width;
#pragma unroll
            for (int f = lane; f < width * kTileK; f += 64) {
                const int kl = f / width, c = f - kl * width;
                const int64_t row = off_s[el] + k0 + kl;
                if (kl < r && row < g.capacity) dst[base + f] = tile[c * (kTileStride + kPad) + kl * 65 + el];
            }
        }
    };
    // ---- dataX = s_i ------------------------------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) stage(c, j, o[c][j]);
    __syncthreads();
    flush(g.X, std::integral_constant<int, D>{});
    __syncthreads();
    // ---- dataZ = s_{i+1} - s_i (data_manipulation.py:84-85) ---------------------------------------------------
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) stage(c, j, o2[c][j] - o[c][j]);
    __syncthreads();
    flush(g.Z, std::integral_constant<int, D>{});
    __syncthreads();
    // ---- dataY = a_i ----------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 16; ++j) stage(0, j, a[j]);
    __syncthreads();
    flush(g.Y, std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------------------
// Column statistics.  A block runs A = (256 / cols) * cols active threads; thread g (global index over the active
// threads) walks elements g, g + S, g + 2S, ... of the flattened matrix with S = gridDim.x * A, a multiple of cols,
// so its column g % cols = threadIdx.x % cols is fixed.  Four independent loads are in flight per thread.  The
// block sums each residue class with a fixed tree, the finalize kernel sums the block partials with a fixed
// pattern: the result does not depend on scheduling.  PASS 0: sum x; PASS 1: sum (x - mean)^2.
constexpr int kStatBlocks = 2048;
template <int PASS>
__global__ __launch_bounds__(kBlock) void column_partials_kernel(const float *__restrict__ x, int64_t total, int32_t cols,
                                                                  const double *__restrict__ mean,
                                                                  double *__restrict__ partial /* [gridDim.x][cols] */) {
    __shared__ double sh[kBlock];
    const int A = (kBlock / cols) * cols;
    const int t = threadIdx.x;
    const int64_t S = (int64_t)gridDim.x * A;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (t < A) {
        const double m = PASS ? mean[t % cols] : 0.0;
        auto f = [&](float v) { const double d = (double)v - m; return PASS ? d * d : d; };
        int64_t e = (int64_t)blockIdx.x * A + t;
        for (; e + 3 * S < total; e += 4 * S) {
            const float v0 = x[e], v1 = x[e + S], v2 = x[e + 2 * S], v3 = x[e + 3 * S];
            a0 += f(v0); a1 += f(v1); a2 += f(v2); a3 += f(v3);
        }
        for (; e < total; e += S) a0 += f(x[e]);
    }
    sh[t] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    const int q = t / cols, cnt = A / cols;       // class r = t % cols holds threads r, r + cols, ... (cnt of them)
    for (int h = 128; h >= 1; h >>= 1) {
        if (t < A && q < h && q + h < cnt) sh[t] += sh[t + cols * h];
        __syncthreads();
    }
    if (t < cols) partial[(int64_t)blockIdx.x * cols + t] = sh[t];
}

// one wave per column: lane l adds blocks l, l + 64, ... in order, then a fixed butterfly
template <int PASS>
__global__ __launch_bounds__(64) void column_finalize_kernel(const double *__restrict__ partial, int nblocks, int32_t cols,
                                                              int64_t rows, double *__restrict__ out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double a = 0.0;
    for (int b = lane; b < nblocks; b += 64) a += partial[(int64_t)b * cols + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    a /= (double)rows;
    if (lane == 0) out[c] = PASS ? sqrt(a) : a;
}

constexpr int kPerThread = 4;     // elements per thread of the elementwise kernels (independent loads in flight)

// Element e of the flattened matrix -> (row, column) without a 64-bit division per element: a block owns the
// contiguous span [blockIdx.x * kBlock * kPerThread, ...), whose first (row, column) is computed once.
struct SpanIndex {
    int64_t row0;
    uint32_t col0;
    __device__ SpanIndex(int64_t first, int32_t cols) : row0(first / cols), col0((uint32_t)(first - (first / cols) * cols)) {}
    __device__ void at(uint32_t local, uint32_t cols, int64_t &r, int &c) const {
        const uint32_t v = col0 + local, q = v / cols;
        r = row0 + q;
        c = (int)(v - q * cols);
    }
};

// COLS > 0: the column count as a compile-time constant (the (row, column) of an element is then a multiply-shift, not
// an integer division per element); COLS == 0: any count
template <int COLS>
__global__ __launch_bounds__(kBlock) void zscore_kernel(const float *__restrict__ x, int64_t total, int32_t cols_rt,
                                                         const double *__restrict__ mean, const double *__restrict__ sd,
                                                         float *__restrict__ out, int32_t out_stride, int32_t out_col0) {
    const int32_t cols = COLS > 0 ? COLS : cols_rt;
    // (x - mean) * (1 / std): one f64 division per column and block instead of one per element; 0 * inf = NaN and
    // finite * inf = +-inf reproduce what the division by a zero std gives
    __shared__ double mean_s[kBlock], inv_s[kBlock];
    if ((int)threadIdx.x < cols) {
        mean_s[threadIdx.x] = mean[threadIdx.x];
        inv_s[threadIdx.x] = 1.0 / sd[threadIdx.x];
    }
    const int64_t first = (int64_t)blockIdx.x * (kBlock * kPerThread);
    const SpanIndex span(first, cols);
    float v[kPerThread];
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const int64_t e = first + threadIdx.x + j * kBlock;
        v[j] = e < total ? x[e] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const uint32_t local = threadIdx.x + j * kBlock;
        if (first + local >= total) break;
        int64_t r;
        int c;
        span.at(local, (uint32_t)cols, r, c);
        double z = ((double)v[j] - mean_s[c]) * inv_s[c];
        // np.nan_to_num (NND_MB_agent.py:305,310,315): NaN -> 0, +-Inf -> +-largest finite (fp32 here)
        if (z != z) z = 0.0;
        float f = (float)z;
        if (f > 3.402823466e38f) f = 3.402823466e38f;
        if (f < -3.402823466e38f) f = -3.402823466e38f;
        out[r * out_stride + out_col0 + c] = f;
    }
}

// The network-input matrix of NND_MB_agent.py:318 in ONE pass: out[r] = [zscore(x[r]) | zscore(y[r])], rows of
// CX + CY floats written back to back.  Two ssc_zscore launches into the same matrix each write 8 (or 4) of every 12
// bytes -- partial lines on the write side, 2.8-3.0 TB/s; here every output line is written whole (3.5 TB/s for 2 + 1
// columns, 4.8 TB/s for 3 + 1; a thread-per-row variant with the constants in registers measured 3.6 / 3.9).
template <int CX, int CY>
__global__ __launch_bounds__(kBlock) void zscore_concat_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                int64_t total, int32_t cx_rt, int32_t cy_rt,
                                                                const double *__restrict__ mean_x, const double *__restrict__ sd_x,
                                                                const double *__restrict__ mean_y, const double *__restrict__ sd_y,
                                                                float *__restrict__ out) {
    const int32_t cx = CX > 0 ? CX : cx_rt, cy = CX > 0 ? CY : cy_rt, cols = cx + cy;
    __shared__ double mean_s[kBlock], inv_s[kBlock];
    if ((int)threadIdx.x < cols) {
        const int c = threadIdx.x;
        mean_s[c] = c < cx ? mean_x[c] : mean_y[c - cx];
        inv_s[c] = 1.0 / (c < cx ? sd_x[c] : sd_y[c - cx]);
    }
    const int64_t first = (int64_t)blockIdx.x * (kBlock * kPerThread);
    const SpanIndex span(first, cols);
    float v[kPerThread];
    int col[kPerThread];
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const uint32_t local = threadIdx.x + j * kBlock;
        int64_t r;
        span.at(local, (uint32_t)cols, r, col[j]);
        const int c = col[j];
        v[j] = first + local < total ? (c < cx ? x[r * cx + c] : y[r * cy + (c - cx)]) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        const int64_t e = first + threadIdx.x + j * kBlock;
        if (e >= total) break;
        double z = ((double)v[j] - mean_s[col[j]]) * inv_s[col[j]];
        if (z != z) z = 0.0;                      // np.nan_to_num, as in zscore_kernel
        float f = (float)z;
        if (f > 3.402823466e38f) f = 3.402823466e38f;
        if (f < -3.402823466e38f) f = -3.402823466e38f;
        out[e] = f;
    }
}

// One thread per row and group of four columns: ONE Philox evaluation (counter stream_id << 8 | group) yields the four
// gaussians of the group -- word pair (x, y) for columns 4g and 4g + 1, (z, w) for 4g + 2 and 4g + 3, the cos output of a
// pair's Box-Muller transform for the even and the sin output for the odd column (oracle: add_noise_keyed).  The first
// version drew one Philox + one Box-Muller per ELEMENT and was bound by that arithmetic at 2.4-3.0 TB/s.
template <int COLS>
__global__ __launch_bounds__(kBlock) void add_noise_kernel(float *__restrict__ x, int64_t rows, int32_t cols_rt,
                                                            const double *__restrict__ mean, double nts, uint64_t seed,
                                                            uint64_t stream_id) {
    const int32_t cols = COLS > 0 ? COLS : cols_rt;
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= rows) return;
    float *row = x + r * cols;
    if (COLS > 0) {
        float v[COLS > 0 ? COLS : 1];
#pragma unroll
        for (int c = 0; c < COLS; ++c) v[c] = row[c];
        const u32x4 wds = rng_words(seed, (uint64_t)r, stream_id << 8, TAG_DATA_NOISE);
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const double sd = mean[c] * nts;
            if (!(sd > 0.0)) continue;                // helper_funcs.py:14: only where mean * noiseToSignal > 0
            row[c] = fmaf((float)sd, gaussian_f32(c < 2 ? wds.x : wds.z, c < 2 ? wds.y : wds.w, (c & 1) != 0), v[c]);
        }
        return;
    }
    for (int g = 0; g * 4 < cols; ++g) {
        const u32x4 wds = rng_words(seed, (uint64_t)r, (stream_id << 8) | (uint64_t)g, TAG_DATA_NOISE);
        for (int j = 0; j < 4 && g * 4 + j < cols; ++j) {
            const int c = g * 4 + j;
            const double sd = mean[c] * nts;
            if (!(sd > 0.0)) continue;
            row[c] = fmaf((float)sd, gaussian_f32(j < 2 ? wds.x : wds.z, j < 2 ? wds.y : wds.w, (j & 1) != 0), row[c]);
        }
    }
}

}  // namespace ssc

using namespace ssc;

extern "C" {

int ssc_dataset_scan(const ssc_transition_log *log, int32_t K, int64_t n, int32_t *d_len, int64_t *d_off,
                     void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    SSC_REQUIRE(log != nullptr, "ssc_dataset_scan: log NULL");
    SSC_REQUIRE(K >= 0 && n >= 0, "ssc_dataset_scan: negative size");
    SSC_REQUIRE(d_off != nullptr, "ssc_dataset_scan: d_off NULL");
    const int64_t groups = (n + kGroup - 1) / kGroup;
    if (n == 0) return check_hip(hipMemsetAsync(d_off, 0, sizeof(int64_t), as_stream(stream)), "ssc_dataset_scan(memset)");
    SSC_REQUIRE(d_len != nullptr && (K == 0 || log->done != nullptr), "ssc_dataset_scan: NULL pointer");
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= ssc_dataset_scan_workspace_bytes(n),
                "ssc_dataset_scan: workspace too small");
    SSC_REQUIRE(groups <= 0x7fffffffLL, "ssc_dataset_scan: too many envs");
    const int64_t drs = log->done_row_stride ? log->done_row_stride : n;
    int64_t *group_rows = static_cast<int64_t *>(d_workspace);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(dataset_len_kernel, dim3((unsigned)groups), dim3(256), 0, s, log->done, drs, K, n, d_len, group_rows);
    hipLaunchKernelGGL(dataset_group_scan_kernel, dim3(1), dim3(kScanThreads), 0, s, group_rows, groups, d_off + n);
    hipLaunchKernelGGL(dataset_offsets_kernel, dim3((unsigned)groups), dim3(64), 0, s, d_len, group_rows, n, d_off);
    return check_launch("ssc_dataset_scan");
}

size_t ssc_dataset_scan_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    return (size_t)((n + kGroup - 1) / kGroup) * sizeof(int64_t) + sizeof(int64_t);
}

int ssc_dataset_build(const ssc_transition_log *log, int32_t obs_dim, int32_t K, int64_t n, const int32_t *d_len,
                      const int64_t *d_off, int64_t capacity_rows, float *d_X, float *d_Y, float *d_Z,
                      ssc_stream_t stream) {
    SSC_REQUIRE(log != nullptr, "ssc_dataset_build: log NULL");
    SSC_REQUIRE(obs_dim >= 1 && obs_dim <= SSC_MAX_OBS, "ssc_dataset_build: obs_dim %d", obs_dim);
    SSC_REQUIRE(K >= 0 && n >= 0 && capacity_rows >= 0, "ssc_dataset_build: negative size");
    if (n == 0 || K <= 1 || capacity_rows == 0) return SSC_OK;
    SSC_REQUIRE(d_len && d_off && d_X && d_Y && d_Z, "ssc_dataset_build: NULL pointer");
    SSC_REQUIRE(log->act != nullptr, "ssc_dataset_build: NULL act column");
    for (int c = 0; c < obs_dim; ++c) SSC_REQUIRE(log->obs[c] && log->obs2[c], "ssc_dataset_build: NULL obs column %d", c);
    DatasetArgs g;
    g.log = *log; g.len = d_len; g.off = d_off; g.X = d_X; g.Y = d_Y; g.Z = d_Z;
    g.n = n; g.rs = log->row_stride ? log->row_stride : n; g.capacity = capacity_rows;
    const int64_t gx = (n + kGroup - 1) / kGroup, gy = (K - 1 + kTileK - 1) / kTileK;
    SSC_REQUIRE(gx <= 0x7fffffffLL && gy <= 65535, "ssc_dataset_build: grid too large");
    const dim3 grid((unsigned)gx, (unsigned)gy);
    static_assert(SSC_MAX_OBS == 3, "dataset_build_kernel is instantiated for obs_dim 1..3");
    if (obs_dim == 1) hipLaunchKernelGGL(dataset_build_kernel<1>, grid, dim3(256), 0, as_stream(stream), g);
    else if (obs_dim == 2) hipLaunchKernelGGL(dataset_build_kernel<2>, grid, dim3(256), 0, as_stream(stream), g);
    else hipLaunchKernelGGL(dataset_build_kernel<3>, grid, dim3(256), 0, as_stream(stream), g);
    return check_launch("ssc_dataset_build");
}

size_t ssc_column_stats_workspace_bytes(int32_t cols) {
    if (cols < 1 || cols > 64) return 0;
    return (size_t)kStatBlocks * (size_t)cols * sizeof(double);
}

int ssc_column_stats(const float *d_x, int64_t rows, int32_t cols, double *d_mean, double *d_std, void *d_workspace,
                     size_t workspace_bytes, ssc_stream_t stream) {
    SSC_REQUIRE(cols >= 1 && cols <= 64, "ssc_column_stats: cols %d not in 1..64", cols);
    SSC_REQUIRE(rows >= 1, "ssc_column_stats: need at least one row (np.mean of an empty array is NaN)");
    SSC_REQUIRE(d_x && d_mean && d_std && d_workspace, "ssc_column_stats: NULL pointer");
    SSC_REQUIRE(workspace_bytes >= ssc_column_stats_workspace_bytes(cols), "ssc_column_stats: workspace too small");
    const int64_t total = rows * cols;
    int64_t nb = (total + (int64_t)kBlock * 16 - 1) / ((int64_t)kBlock * 16);
    const int blocks = (int)(nb < 1 ? 1 : nb > kStatBlocks ? kStatBlocks : nb);
    double *part = static_cast<double *>(d_workspace);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(column_partials_kernel<0>, dim3(blocks), dim3(kBlock), 0, s, d_x, total, cols, nullptr, part);
    hipLaunchKernelGGL(column_finalize_kernel<0>, dim3(cols), dim3(64), 0, s, part, blocks, cols, rows, d_mean);
    hipLaunchKernelGGL(column_partials_kernel<1>, dim3(blocks), dim3(kBlock), 0, s, d_x, total, cols, d_mean, part);
    hipLaunchKernelGGL(column_finalize_kernel<1>, dim3(cols), dim3(64), 0, s, part, blocks, cols, rows, d_std);
    return check_launch("ssc_column_stats");
}

int ssc_zscore(const float *d_x, int64_t rows, int32_t cols, const double *d_mean, const double *d_std, float *d_out,
               int32_t out_stride, int32_t out_col0, ssc_stream_t stream) {
    SSC_REQUIRE(cols >= 1 && cols <= kBlock && rows >= 0 && out_col0 >= 0 && out_stride >= out_col0 + cols,
                "ssc_zscore: cols %d, out_stride %d, out_col0 %d", cols, out_stride, out_col0);
    if (rows == 0) return SSC_OK;
    SSC_REQUIRE(d_x && d_mean && d_std && d_out, "ssc_zscore: NULL pointer");
    const int64_t total = rows * cols;
    const dim3 grid(blocks_for(total, kBlock * kPerThread)), block(kBlock);
    hipStream_t s = as_stream(stream);
#define SSC_ZS(C) hipLaunchKernelGGL(zscore_kernel<C>, grid, block, 0, s, d_x, total, cols, d_mean, d_std, d_out, out_stride, out_col0)
    if (cols == 1) SSC_ZS(1); else if (cols == 2) SSC_ZS(2); else if (cols == 3) SSC_ZS(3); else if (cols == 4) SSC_ZS(4); else SSC_ZS(0);
#undef SSC_ZS
    return check_launch("ssc_zscore");
}

int ssc_zscore_concat(const float *d_x, int32_t cols_x, const double *d_mean_x, const double *d_std_x, const float *d_y,
                      int32_t cols_y, const double *d_mean_y, const double *d_std_y, int64_t rows, float *d_out,
                      ssc_stream_t stream) {
    SSC_REQUIRE(cols_x >= 1 && cols_y >= 1 && cols_x + cols_y <= kBlock && rows >= 0, "ssc_zscore_concat: cols %d + %d, rows %lld",
                cols_x, cols_y, (long long)rows);
    if (rows == 0) return SSC_OK;
    SSC_REQUIRE(d_x && d_y && d_mean_x && d_std_x && d_mean_y && d_std_y && d_out, "ssc_zscore_concat: NULL pointer");
    const int64_t total = rows * (cols_x + cols_y);
    const dim3 grid(blocks_for(total, kBlock * kPerThread)), block(kBlock);
    hipStream_t s = as_stream(stream);
#define SSC_ZC(A, B) hipLaunchKernelGGL((zscore_concat_kernel<A, B>), grid, block, 0, s, d_x, d_y, total, cols_x, cols_y, d_mean_x, d_std_x, d_mean_y, d_std_y, d_out)
    if (cols_x == 2 && cols_y == 1) SSC_ZC(2, 1); else if (cols_x == 3 && cols_y == 1) SSC_ZC(3, 1); else SSC_ZC(0, 0);
#undef SSC_ZC
    return check_launch("ssc_zscore_concat");
}

int ssc_add_noise(float *d_x, int64_t rows, int32_t cols, const double *d_mean, double noise_to_signal, uint64_t seed,
                  uint64_t stream_id, ssc_stream_t stream) {
    SSC_REQUIRE(cols >= 1 && cols <= 256 && rows >= 0, "ssc_add_noise: cols %d not in 1..256", cols);
    SSC_REQUIRE(stream_id < (1ull << 48), "ssc_add_noise: stream_id too large");
    if (rows == 0) return SSC_OK;
    SSC_REQUIRE(d_x && d_mean, "ssc_add_noise: NULL pointer");
    const dim3 grid(blocks_for(rows, kBlock)), block(kBlock);
    hipStream_t s = as_stream(stream);
#define SSC_AN(C) hipLaunchKernelGGL(add_noise_kernel<C>, grid, block, 0, s, d_x, rows, cols, d_mean, noise_to_signal, seed, stream_id)
    if (cols == 1) SSC_AN(1); else if (cols == 2) SSC_AN(2); else if (cols == 3) SSC_AN(3); else if (cols == 4) SSC_AN(4); else SSC_AN(0);
#undef SSC_AN
    return check_launch("ssc_add_noise");
}

}  // extern "C"
