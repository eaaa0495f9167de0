// dyn_train.hip -- one Adam step of the NND_MB dynamics model on a mini-batch
// (NN_Dynamics_Model/dynamics_model.py:41-50, 98-113) for ANY feedforward_network shape.
//
// fp32 throughout (the reference trains in fp64 on the CPU; parity tolerance in the tests).
//
// Two paths behind ssc_mlp_train_steps:
//   * mlp_train_fused_kernel -- ONE launch per step for the shapes the reference ships (one hidden layer of
//     <= 512 units: 1x32 in the examples, 1x500 the class default).  Block b owns 32 batch rows: gather, forward,
//     output delta, back-propagation and its share of every gradient, all in LDS/registers; the block that
//     finishes last (atomic ticket) sums the per-block gradients in block order and applies Adam.
//   * any other feedforward_network shape: gather -> per layer a forward GEMM (the output layer's epilogue leaves the
//     output delta and per-tile loss sums) -> per hidden layer a backward-data GEMM with the ReLU mask -> ONE launch
//     with the weight-gradient GEMMs of all layers, Adam in the epilogue; all on the exact-fp32 MFMA through one
//     32x32-tile routine (gemm32_tile).  2 L + 1 launches per step.
// The bias-corrected step size comes from a device-side step counter, so any number of consecutive steps is
// enqueued by one call without a host round trip.
#include <stdlib.h>

#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

// batch rows by index; block 0 also advances the Adam step counter: scal[0] = lr_t of this step
// (tf.train.AdamOptimizer: lr * sqrt(1 - b2^t) / (1 - b1^t))
__global__ __launch_bounds__(256) void train_gather_kernel(int B, int in, int out, const float *__restrict__ X,
                                                           const float *__restrict__ Z, const int32_t *__restrict__ idx,
                                                           float *__restrict__ xb, float *__restrict__ zb, int32_t *t,
                                                           float lr, float b1, float b2, float *scal) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int tt = t[0] + 1;
        t[0] = tt;
        scal[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)tt)) / (1.0 - pow((double)b1, (double)tt)));
    }
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < B * in) {
        const int r = e / in, c = e - r * in;
        xb[e] = X[(int64_t)idx[r] * in + c];
    }
    if (e < B * out) {
        const int r = e / out, c = e - r * out;
        zb[e] = Z[(int64_t)idx[r] * out + c];
    }
}

// ---------------------------------------------------------------------------------------------------------
// The three GEMMs of a layer on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32):
//   forward   Y  = act(X W + b)         A = X  [B x K] rows,      B = W  [K x N] rows
//   backward  dX = (dY W^T) * (X > 0)   A = dY [B x N] rows,      B = W^T: B[k][n] = W[n][k]
//   gradient  dW = X^T dY (+ ones row -> db), Adam in the epilogue:  A = X^T: A[m][k] = X[k][m],  B = dY rows
// One block = one 32 x 32 output tile, its K range split over the 4 waves (one per SIMD) and summed through LDS in
// wave order.  Operands go straight from global memory / L2 to the MFMA registers: a "row" operand is read as one
// 16-byte vector per lane (lane = output row or column, 4 consecutive k), a "column" operand as 4 coalesced dwords.
// Lanes 0-31 feed k = 8q + r, lanes 32-63 k = 8q + 4 + r into the r-th MFMA of group q -- any pairing of k values
// is a valid contraction as long as A and B agree on it.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_ADAM = 3, EPI_DELTA = 4 };

struct GemmArgs {
    const float *A, *B;
    int lda, ldb, M, N, K;
    int a_vec, b_vec;              // 16-byte loads allowed for a row operand (alignment + K % 4 == 0)
    int ones_row;                  // gradient GEMM: A row that reads as 1.0 (bias gradient), else -1
    float *C; int ldc;             // EPI_BIAS / EPI_BIAS_RELU / EPI_MASK
    const float *bias;             // EPI_BIAS*
    const float *mask; int ldmask; // EPI_MASK
    const float *z; float delta_scale; float *loss_part;   // EPI_DELTA: C = (acc + bias - z) * delta_scale (the output
                                   // delta, d mean((z - y)^2) / dy, dynamics_model.py:41); loss_part[tile] = sum (y - z)^2
    float *W, *mW, *vW, *bb, *mb, *vb;   // EPI_ADAM: W [M - 1 or M][N], bias [N]
    const float *scal;             // scal[0] = lr_t
    float beta1, beta2, eps;
};

// 4 k-values of operand element `idx` (row or column of the tile) for this lane's half
template <bool ROWS>
__device__ __forceinline__ void gemm_load4(const float *__restrict__ P, int ld, int idx, int lim, int k, int K, int vec,
                                           int ones_idx, float (&out)[4]) {
    if (ROWS) {      // P[idx * ld + k .. k + 3]
        const float *src = P + (int64_t)idx * ld + k;
        if (idx < lim && vec && k + 3 < K) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = (idx < lim && k + r < K) ? src[r] : 0.0f;
        }
    } else {         // P[(k + r) * ld + idx]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.0f;
            if (k + r < K) {
                if (idx == ones_idx) v = 1.0f;
                else if (idx < lim) v = P[(int64_t)(k + r) * ld + idx];
            }
            out[r] = v;
        }
    }
}

template <bool A_ROWS, bool B_ROWS, int EPI>
__device__ __forceinline__ void gemm32_tile(const GemmArgs &g, float (&red)[4][16][64]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int ksz = (((g.K + 3) >> 2) + 7) & ~7;          // k values per wave, a multiple of 8
    const int kb = w * ksz, ke = kb + ksz < g.K ? kb + ksz : g.K;
    const int a_lim = g.ones_row >= 0 ? g.ones_row : g.M;  // the ones row is not read from memory
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // Chunks of 4 k-groups (32 k values), double-buffered: the operands of chunk c + 1 are requested before the 16
    // MFMAs of chunk c, so a chunk costs max(load latency, MFMA time) instead of one load latency per k-group.
    constexpr int CH = 4;
    float a[2][CH][4], b[2][CH][4];
    auto load_chunk = [&](int buf, int k0) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {      // past ke the loaders return zeros
            gemm_load4<A_ROWS>(g.A, g.lda, m0 + l32, a_lim, k0 + 8 * u + 4 * half, ke, g.a_vec, A_ROWS ? -1 : g.ones_row,
                               a[buf][u]);
            gemm_load4<B_ROWS>(g.B, g.ldb, n0 + l32, g.N, k0 + 8 * u + 4 * half, ke, g.b_vec, -1, b[buf][u]);
        }
    };
    auto mfma_chunk = [&](int buf) {
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][u][r], b[buf][u][r], acc, 0, 0, 0);
    };
    if (kb < ke) load_chunk(0, kb);
    for (int k0 = kb; k0 < ke; k0 += 16 * CH) {      // two chunks per trip: static buffer indices
        if (k0 + 8 * CH < ke) load_chunk(1, k0 + 8 * CH);
        mfma_chunk(0);
        if (k0 + 8 * CH < ke) {
            if (k0 + 16 * CH < ke) load_chunk(0, k0 + 16 * CH);
            mfma_chunk(1);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[w][i][lane] = acc[i];
    __syncthreads();
    // wave w finishes accumulator registers 4w .. 4w + 3: rows m0 + 8w + 4 half + (i & 3), column n0 + l32
    const int n = n0 + l32;
    float lsum = 0.0f;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int i = 4 * w + ii;
        const float v = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
        const int m = m0 + 8 * w + 4 * half + ii;
        if (m >= g.M || n >= g.N) continue;
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
            const float y = v + g.bias[n];
            g.C[(int64_t)m * g.ldc + n] = EPI == EPI_BIAS_RELU ? fmaxf(y, 0.0f) : y;
        } else if (EPI == EPI_MASK) {
            g.C[(int64_t)m * g.ldc + n] = g.mask[(int64_t)m * g.ldmask + n] > 0.0f ? v : 0.0f;
        } else if (EPI == EPI_DELTA) {
            const float d = v + g.bias[n] - g.z[(int64_t)m * g.ldc + n];
            g.C[(int64_t)m * g.ldc + n] = d * g.delta_scale;
            lsum = fmaf(d, d, lsum);
        } else {
            float *th, *mm, *vv;
            if (m == g.ones_row) { th = g.bb + n; mm = g.mb + n; vv = g.vb + n; }
            else { const int64_t e = (int64_t)m * g.N + n; th = g.W + e; mm = g.mW + e; vv = g.vW + e; }
            const float m1 = g.beta1 * *mm + (1.0f - g.beta1) * v;
            const float v1 = g.beta2 * *vv + (1.0f - g.beta2) * v * v;
            *mm = m1; *vv = v1;
            *th -= g.scal[0] * m1 / (sqrtf(v1) + g.eps);
        }
    }
    if (EPI == EPI_DELTA) {      // sum (y - z)^2 of this tile, fixed order: lanes (butterfly), then waves 0..3
        __shared__ float lpart[4];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o);
        if (lane == 0) lpart[w] = lsum;
        __syncthreads();
        if (threadIdx.x == 0)
            g.loss_part[blockIdx.y * gridDim.x + blockIdx.x] = ((lpart[0] + lpart[1]) + lpart[2]) + lpart[3];
    }
}

template <bool A_ROWS, bool B_ROWS, int EPI>
__global__ __launch_bounds__(256) void gemm32_f32_kernel(GemmArgs g) {
    __shared__ float red[4][16][64];
    gemm32_tile<A_ROWS, B_ROWS, EPI>(g, red);
}

// the weight-gradient GEMMs of ALL layers in one launch (they are independent once every delta exists):
// blockIdx.z picks the layer, tiles outside its matrix leave at once
struct GemmBatch {
    GemmArgs p[SSC_MAX_LAYERS];
    const float *loss_part;      // per-tile sums of (y - z)^2 left by the EPI_DELTA GEMM
    int n_loss_part;
    float loss_scale;            // 1 / (B * out)
    float *loss;                 // [1] or NULL
};
__global__ __launch_bounds__(256) void gemm32_wgrad_batch_kernel(GemmBatch b) {
    __shared__ float red[4][16][64];
    if (b.loss != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        float a = 0.0f;                                    // the batch MSE, tiles summed in order
        for (int i = 0; i < b.n_loss_part; ++i) a += b.loss_part[i];
        b.loss[0] = a * b.loss_scale;
    }
    const GemmArgs &g = b.p[blockIdx.z];
    if ((int)blockIdx.x * 32 >= g.N || (int)blockIdx.y * 32 >= g.M) return;
    gemm32_tile<false, false, EPI_ADAM>(g, red);
}

static bool vec_ok(const float *p, int ld, int K) {
    return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % 4 == 0) && (K % 4 == 0);
}

template <bool A_ROWS, bool B_ROWS, int EPI>
static void launch_gemm(const GemmArgs &g, hipStream_t s) {
    hipLaunchKernelGGL((gemm32_f32_kernel<A_ROWS, B_ROWS, EPI>), dim3((g.N + 31) / 32, (g.M + 31) / 32), dim3(256), 0, s, g);
}

// ---------------------------------------------------------------------------------------------------------
// Fused step for one hidden layer:  y = relu(x W1 + b1) W2 + b2.
constexpr int kFT = 512;      // threads per block
constexpr int kFRows = 32;    // batch rows per block

// tf.train.AdamOptimizer's bias-corrected step size lr * sqrt(1 - beta2^t) / (1 - beta1^t): the powers are
// computed once per call (pow) and advanced by one multiplication per step
struct FusedStepState {
    double pw1, pw2;     // beta1^t, beta2^t for the COMING step t
    float lr_t;
};

struct FusedTrainArgs {
    const float *X, *Z;
    const int32_t *idx;
    int B, in, hd, out, hd_pad, hd_shift, S, G;
    float *W1, *b1, *W2, *b2;
    float *mW1, *vW1, *mb1, *vb1, *mW2, *vW2, *mb2, *vb2;
    int32_t *adam_t;
    float lr, beta1, beta2, eps;
    float *partial;      // [G][P + 1]: per-block gradients in flat parameter order (W1 | b1 | W2 | b2), then sum (y-z)^2
    uint32_t *ticket;    // [1], zero before the first launch; the last block leaves it zero again
    FusedStepState *state;   // beta powers and the step size of the coming step (fused_begin_kernel)
    float *loss;         // [1] or NULL
};

static inline size_t fused_lds_floats(int hd_pad, int S, int IN, int OUT) {
    return (size_t)kFRows * (hd_pad + 1) + (size_t)hd_pad * OUT + (size_t)kFRows * IN + 2 * (size_t)kFRows * OUT +
           (S > 1 ? (size_t)kFT * (IN + 1 + OUT) : 0) + kFRows + 8;
}

template <int IN, int OUT>
__global__ __launch_bounds__(kFT) void mlp_train_fused_kernel(FusedTrainArgs g) {
    extern __shared__ float lds[];
    float *h_s = lds;                                   // [32][hd_pad + 1] hidden activations of the block's rows
    float *w2_s = h_s + kFRows * (g.hd_pad + 1);        // [hd_pad][OUT]
    float *x_s = w2_s + g.hd_pad * OUT;                 // [32][IN]
    float *z_s = x_s + kFRows * IN;                     // [32][OUT]
    float *dy_s = z_s + kFRows * OUT;                   // [32][OUT]
    float *sc_s = dy_s + kFRows * OUT;                  // [IN + 1 + OUT][kFT] per-thread gradient partials (S > 1)
    float *loss_s = sc_s + (g.S > 1 ? kFT * (IN + 1 + OUT) : 0);   // [32]
    float *misc_s = loss_s + kFRows;                    // [0] last-block flag, [1] lr_t
    const int t = threadIdx.x;
    const int u = t & (g.hd_pad - 1), s = t >> g.hd_shift;          // hidden unit, row sub-slice
    const bool unit_ok = u < g.hd;
    const int row0 = blockIdx.x * kFRows;
    const int nrows = g.B - row0 < kFRows ? g.B - row0 : kFRows;
    const int hp = g.hd_pad + 1;

    // this thread's unit: column u of W1, b1[u], row u of W2 (registers)
    // (every load unconditional -- index clamped into the array, value dropped by a select: a guarded load compiles to a
    // branch with a full wait behind it, and this prologue was seven round trips in a row; the batch's record numbers go
    // out first, their rows are the only dependent request)
    const bool live = t < nrows;
    const int64_t src = g.idx[row0 + (live ? t : 0)];          // (threads >= kFRows read a valid entry too and drop it)
    const int uc = min(u, g.hd - 1);
    float w1[IN], w2[OUT];
    const float xb1 = g.b1[uc];
#pragma unroll
    for (int i = 0; i < IN; ++i) w1[i] = g.W1[min(i, g.in - 1) * g.hd + uc];
#pragma unroll
    for (int o = 0; o < OUT; ++o) w2[o] = g.W2[uc * g.out + min(o, g.out - 1)];
    float xr[IN], zr[OUT];
#pragma unroll
    for (int i = 0; i < IN; ++i) xr[i] = g.X[src * g.in + min(i, g.in - 1)];
#pragma unroll
    for (int o = 0; o < OUT; ++o) zr[o] = g.Z[src * g.out + min(o, g.out - 1)];
    const float bb1 = unit_ok ? xb1 : 0.0f;
#pragma unroll
    for (int i = 0; i < IN; ++i) w1[i] = (unit_ok && i < g.in) ? w1[i] : 0.0f;
#pragma unroll
    for (int o = 0; o < OUT; ++o) w2[o] = (unit_ok && o < g.out) ? w2[o] : 0.0f;
    if (s == 0) {
#pragma unroll
        for (int o = 0; o < OUT; ++o) w2_s[u * OUT + o] = w2[o];
    }
    // batch rows of this block (train_gather_kernel's job)
    if (t < kFRows) {
#pragma unroll
        for (int i = 0; i < IN; ++i) x_s[t * IN + i] = (live && i < g.in) ? xr[i] : 0.0f;
#pragma unroll
        for (int o = 0; o < OUT; ++o) z_s[t * OUT + o] = (live && o < g.out) ? zr[o] : 0.0f;
    }
    __syncthreads();

    // pass 1: h[r][u] = relu(x[r] . W1[:, u] + b1[u])        (feedforward_network.py:14-19)
#pragma unroll 4
    for (int r = s; r < kFRows; r += g.S) {
        float h = bb1;
#pragma unroll
        for (int i = 0; i < IN; ++i) h = fmaf(x_s[r * IN + i], w1[i], h);
        h_s[r * hp + u] = (unit_ok && r < nrows) ? fmaxf(h, 0.0f) : 0.0f;
    }
    __syncthreads();

    // pass 2: y[r][o] = h[r] . W2[:, o] + b2[o]; 16 lanes share a row, lane c takes units c, c + 16, ...
    {
        const int r2 = t >> 4, c = t & 15;
        float acc[OUT];
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = 0.0f;
        for (int uu = c; uu < g.hd_pad; uu += 16) {
            const float hv = h_s[r2 * hp + uu];
#pragma unroll
            for (int o = 0; o < OUT; ++o) acc[o] = fmaf(hv, w2_s[uu * OUT + o], acc[o]);
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
            for (int o = 0; o < OUT; ++o) acc[o] += __shfl_xor(acc[o], m);
        if (c == 0) {
            float lp = 0.0f;
            const float scale = 2.0f / (float)(g.B * g.out);       // d mean((z - y)^2) / dy   (dynamics_model.py:41)
#pragma unroll
            for (int o = 0; o < OUT; ++o) {
                const bool valid = r2 < nrows && o < g.out;
                const float d = valid ? acc[o] + g.b2[o] - z_s[r2 * OUT + o] : 0.0f;
                dy_s[r2 * OUT + o] = d * scale;
                lp = fmaf(d, d, lp);
            }
            loss_s[r2] = lp;
        }
    }
    __syncthreads();

    // pass 3: gradients of this thread's unit over its rows
    float g1[IN], g2[OUT], gb1 = 0.0f;
#pragma unroll
    for (int i = 0; i < IN; ++i) g1[i] = 0.0f;
#pragma unroll
    for (int o = 0; o < OUT; ++o) g2[o] = 0.0f;
#pragma unroll 4
    for (int r = s; r < kFRows; r += g.S) {
        const float hv = h_s[r * hp + u];
        float dh = 0.0f;
#pragma unroll
        for (int o = 0; o < OUT; ++o) {
            const float dyo = dy_s[r * OUT + o];
            g2[o] = fmaf(hv, dyo, g2[o]);
            dh = fmaf(dyo, w2[o], dh);
        }
        dh = hv > 0.0f ? dh : 0.0f;                                // ReLU mask
#pragma unroll
        for (int i = 0; i < IN; ++i) g1[i] = fmaf(x_s[r * IN + i], dh, g1[i]);
        gb1 += dh;
    }
    if (g.S > 1) {       // sum the S row sub-slices of a unit in a fixed order
#pragma unroll
        for (int i = 0; i < IN; ++i) sc_s[i * kFT + t] = g1[i];
        sc_s[IN * kFT + t] = gb1;
#pragma unroll
        for (int o = 0; o < OUT; ++o) sc_s[(IN + 1 + o) * kFT + t] = g2[o];
        __syncthreads();
        if (s == 0) {
            for (int ss = 1; ss < g.S; ++ss) {
                const int tt = u + (ss << g.hd_shift);
#pragma unroll
                for (int i = 0; i < IN; ++i) g1[i] += sc_s[i * kFT + tt];
                gb1 += sc_s[IN * kFT + tt];
#pragma unroll
                for (int o = 0; o < OUT; ++o) g2[o] += sc_s[(IN + 1 + o) * kFT + tt];
            }
        }
    }
    // block partials -> global, flat parameter order
    const int oB1 = g.in * g.hd, oW2 = oB1 + g.hd, oB2 = oW2 + g.hd * g.out, P = oB2 + g.out;
    float *mine = g.partial + (size_t)blockIdx.x * (P + 1);
    // Agent-scope (write-through) stores, waited for before the ticket, and agent-scope loads in the last block: the
    // partials cross XCDs without an L2 write-back / invalidate (a full __threadfence costs more than the step).
    auto put = [&](int p, float v) { __hip_atomic_store(mine + p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (s == 0 && unit_ok) {
#pragma unroll
        for (int i = 0; i < IN; ++i)
            if (i < g.in) put(i * g.hd + u, g1[i]);
        put(oB1 + u, gb1);
#pragma unroll
        for (int o = 0; o < OUT; ++o)
            if (o < g.out) put(oW2 + u * g.out + o, g2[o]);
    }
    if (t < g.out) {     // b2 gradient: sum of dy over the block's rows
        float a = 0.0f;
        for (int r = 0; r < kFRows; ++r) a += dy_s[r * OUT + t];
        put(oB2 + t, a);
    }
    if (t == 0) {
        float a = 0.0f;
        for (int r = 0; r < kFRows; ++r) a += loss_s[r];
        put(P, a);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's partial stores have been acknowledged
    __syncthreads();
    if (t == 0) {
        const uint32_t tk = __hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = tk == (uint32_t)g.G - 1;
        misc_s[0] = last ? 1.0f : 0.0f;
        if (last) {
            __hip_atomic_store(g.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            misc_s[1] = g.state->lr_t;
        }
    }
    __syncthreads();
    if (misc_s[0] == 0.0f) return;
    const float lr_t = misc_s[1];
    if (t == 0) {        // state of the next step (only this block, only this thread touches it)
        const double p1 = g.state->pw1 * (double)g.beta1, p2 = g.state->pw2 * (double)g.beta2;
        g.state->pw1 = p1; g.state->pw2 = p2;
        g.state->lr_t = (float)((double)g.lr * sqrt(1.0 - p2) / (1.0 - p1));
        g.adam_t[0] = g.adam_t[0] + 1;
    }
    auto get = [&](int b, int p) {
        return __hip_atomic_load(g.partial + (size_t)b * (P + 1) + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // Adam on the summed gradients: 4 parameters per thread and round trip (their 4 x 16 block partials and their
    // theta / m / v are requested together), blocks summed in block order
    constexpr int kU = 4;
    for (int p0 = t; p0 < P; p0 += kU * kFT) {
        float *th[kU], *m[kU], *v[kU];
        float th0[kU], m0[kU], v0[kU], gr[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            const int p = p0 + k * kFT;
            const bool ok = p < P;
            int q;
            if (p < oB1) { th[k] = g.W1; m[k] = g.mW1; v[k] = g.vW1; q = p; }
            else if (p < oW2) { th[k] = g.b1; m[k] = g.mb1; v[k] = g.vb1; q = p - oB1; }
            else if (p < oB2) { th[k] = g.W2; m[k] = g.mW2; v[k] = g.vW2; q = p - oW2; }
            else { th[k] = g.b2; m[k] = g.mb2; v[k] = g.vb2; q = ok ? p - oB2 : 0; }
            th[k] += q; m[k] += q; v[k] += q;
            th0[k] = ok ? *th[k] : 0.0f; m0[k] = ok ? *m[k] : 0.0f; v0[k] = ok ? *v[k] : 0.0f;
            gr[k] = 0.0f;
        }
        for (int b0 = 0; b0 < g.G; b0 += 16) {
            float pv[kU][16];
#pragma unroll
            for (int k = 0; k < kU; ++k)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    pv[k][j] = (b0 + j < g.G && p0 + k * kFT < P) ? get(b0 + j, p0 + k * kFT) : 0.0f;
#pragma unroll
            for (int k = 0; k < kU; ++k)
#pragma unroll
                for (int j = 0; j < 16; ++j) gr[k] += pv[k][j];
        }
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            if (p0 + k * kFT >= P) break;
            const float mm = g.beta1 * m0[k] + (1.0f - g.beta1) * gr[k];
            const float vv = g.beta2 * v0[k] + (1.0f - g.beta2) * gr[k] * gr[k];
            *m[k] = mm; *v[k] = vv;
            *th[k] = th0[k] - lr_t * mm / (sqrtf(vv) + g.eps);
        }
    }
    if (t == kFT - 1 && g.loss != nullptr) {
        float a = 0.0f;
        for (int b = 0; b < g.G; ++b) a += get(b, P);
        g.loss[0] = a / (float)(g.B * g.out);
    }
}

// once per ssc_mlp_train_steps call: ticket = 0, beta powers and step size of the first step of the call
__global__ void fused_begin_kernel(const int32_t *adam_t, float lr, float b1, float b2, uint32_t *ticket, FusedStepState *st) {
    const int tt = adam_t[0] + 1;
    const double p1 = pow((double)b1, (double)tt), p2 = pow((double)b2, (double)tt);
    st->pw1 = p1; st->pw2 = p2;
    st->lr_t = (float)((double)lr * sqrt(1.0 - p2) / (1.0 - p1));
    ticket[0] = 0u;
}

static bool fused_eligible(const ssc_mlp_train_desc *net) {
    static const bool off = getenv("SSC_DYN_TRAIN_GENERIC") != nullptr;    // A/B switch for the tests and tools
    return !off && net->n_layers == 2 && net->dims[0] <= 12 && net->dims[1] <= 512 && net->dims[2] <= 8;
}

static int fused_param_count(const ssc_mlp_train_desc *net) {
    return net->dims[0] * net->dims[1] + net->dims[1] + net->dims[1] * net->dims[2] + net->dims[2];
}

template <int IN, int OUT>
static int launch_fused(const FusedTrainArgs &g, hipStream_t s) {
    const size_t lds_bytes = fused_lds_floats(g.hd_pad, g.S, IN, OUT) * sizeof(float);
    // The opt-in is a per-DEVICE attribute of the function: set it on every launch that needs it (like actor.hip,
    // ddpg_train.hip and dyn_mfma.hip do) -- a process-wide "done" flag would skip it on a second device or race
    // between threads.
    if (lds_bytes > 64 * 1024) {
        if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_train_fused_kernel<IN, OUT>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes),
                               "hipFuncSetAttribute(mlp_train_fused_kernel)"))
            return rc;
    }
    hipLaunchKernelGGL((mlp_train_fused_kernel<IN, OUT>), dim3(g.G), dim3(kFT), lds_bytes, s, g);
    return SSC_OK;
}

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ssc

using namespace ssc;

// Dyn_Model.run_validation (dynamics_model.py:173-196): mse_ = reduce_mean(square(z - prediction)) (:42) per batch of
// `batch_elems` = batchsize * out_dim consecutive elements, then the mean over the batches.  One block per batch, the same
// summation order every run.
__global__ __launch_bounds__(256) void mse_batches_kernel(const float *__restrict__ pred, const float *__restrict__ z,
                                                          int64_t batch_elems, float *__restrict__ batch_loss) {
    __shared__ float red[4];
    const float *p = pred + (int64_t)blockIdx.x * batch_elems, *t = z + (int64_t)blockIdx.x * batch_elems;
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < batch_elems; i += 256) { const float e = t[i] - p[i]; s += e * e; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) batch_loss[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)batch_elems;
}

__global__ void mean_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    float s = 0.0f;
    for (int64_t i = 0; i < n; ++i) s += x[i];          // avg_loss += loss, in batch order (dynamics_model.py:188)
    out[0] = s / (float)n;
}

extern "C" {

int ssc_mse_batches(const float *d_pred, const float *d_z, int64_t n_batches, int64_t batch_elems, float *d_batch_loss,
                    float *d_mean, ssc_stream_t stream) {
    SSC_REQUIRE(n_batches >= 1 && n_batches <= 0x7fffffff && batch_elems >= 1, "ssc_mse_batches: n_batches %lld, batch_elems %lld",
                (long long)n_batches, (long long)batch_elems);
    SSC_REQUIRE(d_pred && d_z && d_batch_loss && d_mean, "ssc_mse_batches: NULL device pointer");
    hipLaunchKernelGGL(mse_batches_kernel, dim3((unsigned)n_batches), dim3(256), 0, as_stream(stream), d_pred, d_z, batch_elems, d_batch_loss);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1), 0, as_stream(stream), d_batch_loss, n_batches, d_mean);
    return check_hip(hipGetLastError(), "ssc_mse_batches");
}


size_t ssc_mlp_train_workspace_bytes(const ssc_mlp_train_desc *net, int32_t B) {
    if (net == nullptr || B <= 0 || net->n_layers < 1 || net->n_layers > SSC_MAX_LAYERS) return 0;
    size_t total = 256;  // scalars
    for (int l = 0; l <= net->n_layers; ++l) total += al256((size_t)B * net->dims[l] * 4);  // activations (0 = x batch)
    total += al256((size_t)B * net->dims[net->n_layers] * 4);  // z batch
    for (int l = 1; l <= net->n_layers; ++l) total += al256((size_t)B * net->dims[l] * 4);  // delta of every layer
    total += al256((((size_t)B + 31) / 32) * (((size_t)net->dims[net->n_layers] + 31) / 32) * 4);  // per-tile loss sums
    if (net->n_layers == 2) {                                  // fused path: ticket + per-block gradients
        const size_t G = ((size_t)B + kFRows - 1) / kFRows;
        const size_t fused = 256 + al256(G * ((size_t)fused_param_count(net) + 1) * 4);
        total = fused > total ? fused : total;
    }
    return total;
}

static int generic_step(const ssc_mlp_train_desc *net, const float *d_X, const float *d_Z, const int32_t *d_idx, int32_t B,
                        float *d_loss, void *d_workspace, hipStream_t s) {
    const int L = net->n_layers;
    char *w = static_cast<char *>(d_workspace);
    float *scal = reinterpret_cast<float *>(w); w += 256;
    float *act[SSC_MAX_LAYERS + 1];
    for (int l = 0; l <= L; ++l) { act[l] = reinterpret_cast<float *>(w); w += al256((size_t)B * net->dims[l] * 4); }
    float *zb = reinterpret_cast<float *>(w); w += al256((size_t)B * net->dims[L] * 4);
    float *dz[SSC_MAX_LAYERS + 1];      // dz[l]: delta at the output of layer l - 1 (l = 1..L)
    for (int l = 1; l <= L; ++l) { dz[l] = reinterpret_cast<float *>(w); w += al256((size_t)B * net->dims[l] * 4); }
    float *loss_part = reinterpret_cast<float *>(w);     // [(B/32) x (out/32) tiles]
    const int in = net->dims[0], out = net->dims[L];
    hipLaunchKernelGGL(train_gather_kernel, dim3(blocks_for((int64_t)B * (in > out ? in : out))), dim3(256), 0, s, B, in,
                       out, d_X, d_Z, d_idx, act[0], zb, net->adam_t, net->lr, net->beta1, net->beta2, scal);
    GemmArgs g;
    for (int l = 0; l < L; ++l) {        // act[l + 1] = relu?(act[l] W_l + b_l)      (feedforward_network.py:14-23)
        const int K = net->dims[l], N = net->dims[l + 1];
        g = GemmArgs{};
        g.A = act[l]; g.lda = K; g.B = net->W[l]; g.ldb = N; g.M = B; g.N = N; g.K = K;
        g.a_vec = vec_ok(act[l], K, K); g.ones_row = -1;
        g.C = act[l + 1]; g.ldc = N; g.bias = net->b[l];
        if (l != L - 1) {
            launch_gemm<true, false, EPI_BIAS_RELU>(g, s);
        } else {       // the output layer leaves the output delta and the per-tile loss sums, not y
            g.C = dz[L]; g.z = zb; g.delta_scale = 2.0f / (float)((int64_t)B * out); g.loss_part = loss_part;
            launch_gemm<true, false, EPI_DELTA>(g, s);
        }
    }
    for (int l = L - 1; l >= 1; --l) {   // deltas of all layers first: they need the OLD weights
        const int K = net->dims[l], N = net->dims[l + 1];
        g = GemmArgs{};
        g.A = dz[l + 1]; g.lda = N; g.B = net->W[l]; g.ldb = N; g.M = B; g.N = K; g.K = N;
        g.a_vec = vec_ok(dz[l + 1], N, N); g.b_vec = vec_ok(net->W[l], N, N); g.ones_row = -1;
        g.C = dz[l]; g.ldc = K; g.mask = act[l]; g.ldmask = K;
        launch_gemm<true, true, EPI_MASK>(g, s);
    }
    // dW_l = act[l]^T dz[l + 1] with a row of ones appended (-> db_l), Adam in the epilogue: one launch for all layers
    GemmBatch batch{};
    int gx = 0, gy = 0;
    for (int l = 0; l < L; ++l) {
        const int K = net->dims[l], N = net->dims[l + 1];
        GemmArgs &q = batch.p[l];
        q.A = act[l]; q.lda = K; q.B = dz[l + 1]; q.ldb = N; q.M = K + 1; q.N = N; q.K = B;
        q.ones_row = K;
        q.W = net->W[l]; q.mW = net->mW[l]; q.vW = net->vW[l]; q.bb = net->b[l]; q.mb = net->mb[l]; q.vb = net->vb[l];
        q.scal = scal; q.beta1 = net->beta1; q.beta2 = net->beta2; q.eps = net->epsilon;
        gx = (N + 31) / 32 > gx ? (N + 31) / 32 : gx;
        gy = (K + 1 + 31) / 32 > gy ? (K + 1 + 31) / 32 : gy;
    }
    batch.loss_part = loss_part; batch.n_loss_part = ((B + 31) / 32) * ((out + 31) / 32);
    batch.loss_scale = 1.0f / (float)((int64_t)B * out); batch.loss = d_loss;
    hipLaunchKernelGGL(gemm32_wgrad_batch_kernel, dim3(gx, gy, L), dim3(256), 0, s, batch);
    return SSC_OK;
}

int ssc_mlp_train_steps(const ssc_mlp_train_desc *net, const float *d_X, const float *d_Z, const int32_t *d_idx,
                        int32_t B, int32_t n_steps, float *d_loss, void *d_workspace, size_t workspace_bytes,
                        ssc_stream_t stream) {
    SSC_REQUIRE(net != nullptr, "ssc_mlp_train_steps: net NULL");
    SSC_REQUIRE(net->n_layers >= 1 && net->n_layers <= SSC_MAX_LAYERS, "ssc_mlp_train_steps: bad n_layers");
    SSC_REQUIRE(B >= 1 && B <= 65536, "ssc_mlp_train_steps: batch %d out of range", B);
    SSC_REQUIRE(n_steps >= 0, "ssc_mlp_train_steps: n_steps %d < 0", n_steps);
    for (int l = 0; l <= net->n_layers; ++l)
        SSC_REQUIRE(net->dims[l] >= 1 && net->dims[l] <= 8192, "ssc_mlp_train_steps: bad dims[%d]", l);
    for (int l = 0; l < net->n_layers; ++l)
        SSC_REQUIRE(net->W[l] && net->b[l] && net->mW[l] && net->vW[l] && net->mb[l] && net->vb[l],
                    "ssc_mlp_train_steps: NULL parameter / moment pointer (layer %d)", l);
    if (n_steps == 0) return SSC_OK;
    SSC_REQUIRE(net->adam_t && d_X && d_Z && d_idx, "ssc_mlp_train_steps: NULL pointer");
    const size_t need = ssc_mlp_train_workspace_bytes(net, B);
    SSC_REQUIRE(d_workspace && workspace_bytes >= need, "ssc_mlp_train_steps: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    if (!fused_eligible(net)) {
        for (int k = 0; k < n_steps; ++k)
            if (int rc = generic_step(net, d_X, d_Z, d_idx + (size_t)k * B, B, d_loss ? d_loss + k : nullptr, d_workspace, s))
                return rc;
        return check_launch("ssc_mlp_train_steps");
    }
    FusedTrainArgs g;
    g.X = d_X; g.Z = d_Z; g.B = B;
    g.in = net->dims[0]; g.hd = net->dims[1]; g.out = net->dims[2];
    g.hd_pad = 16; g.hd_shift = 4;
    while (g.hd_pad < g.hd) { g.hd_pad <<= 1; ++g.hd_shift; }
    g.S = kFT / g.hd_pad;
    g.G = (B + kFRows - 1) / kFRows;
    g.W1 = net->W[0]; g.b1 = net->b[0]; g.W2 = net->W[1]; g.b2 = net->b[1];
    g.mW1 = net->mW[0]; g.vW1 = net->vW[0]; g.mb1 = net->mb[0]; g.vb1 = net->vb[0];
    g.mW2 = net->mW[1]; g.vW2 = net->vW[1]; g.mb2 = net->mb[1]; g.vb2 = net->vb[1];
    g.adam_t = net->adam_t; g.lr = net->lr; g.beta1 = net->beta1; g.beta2 = net->beta2; g.eps = net->epsilon;
    g.ticket = static_cast<uint32_t *>(d_workspace);
    g.state = reinterpret_cast<FusedStepState *>(static_cast<char *>(d_workspace) + 64);
    g.partial = reinterpret_cast<float *>(static_cast<char *>(d_workspace) + 256);
    hipLaunchKernelGGL(fused_begin_kernel, dim3(1), dim3(1), 0, s, net->adam_t, net->lr, net->beta1, net->beta2, g.ticket,
                       g.state);
    for (int k = 0; k < n_steps; ++k) {
        g.idx = d_idx + (size_t)k * B;
        g.loss = d_loss ? d_loss + k : nullptr;
        int rc;
        if (g.in <= 3 && g.out <= 2) rc = launch_fused<3, 2>(g, s);
        else if (g.in <= 4 && g.out <= 3) rc = launch_fused<4, 3>(g, s);
        else rc = launch_fused<12, 8>(g, s);
        if (rc) return rc;
    }
    return check_launch("ssc_mlp_train_steps");
}

int ssc_mlp_train_step(const ssc_mlp_train_desc *net, const float *d_X, const float *d_Z, const int32_t *d_idx,
                       int32_t B, float *d_loss, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    return ssc_mlp_train_steps(net, d_X, d_Z, d_idx, B, 1, d_loss, d_workspace, workspace_bytes, stream);
}

}  // extern "C"
