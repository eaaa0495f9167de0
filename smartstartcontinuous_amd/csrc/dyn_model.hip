// dyn_model.hip -- NND_MB dynamics model, fp32 path: feedforward_network
// (NN_Dynamics_Model/feedforward_network.py:3-23) and Dyn_Model.do_forward_sim
// (NN_Dynamics_Model/dynamics_model.py:204-240).
//
// This is the exact-fp32 (VALU) path for ANY layer sizes: the parity reference on the GPU and
// the fallback for shapes the bf16-MFMA kernel (dyn_mfma.hip) does not cover.  One launch per
// layer; activations ping-pong through a caller-provided workspace.
//
// NOTE: compiled WITHOUT -fno-honor-nans -- np.nan_to_num semantics need real NaN/Inf tests.
#include <float.h>

#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

constexpr int kRows = 16;     // rows per block in the layer kernel
constexpr int kKChunk = 512;  // K staged through LDS per pass

// Y[m][N] = act(X[m][K] @ W[K][N] + b).  Block = 256 threads = kRows rows x all N columns
// (thread -> columns tid, tid+256, ...).  The X tile is staged transposed in LDS ([k][row]) so that
// the 16 row values of one k are four broadcast ds_read_b128; W rows are read coalesced from
// L2 (each W element is used for 16 rows).  fp32 FMA chain in k order.
template <bool RELU>
__global__ __launch_bounds__(256) void mlp_layer_f32_kernel(int64_t m, int K, int N, const float *__restrict__ X,
                                                            const float *__restrict__ W,
                                                            const float *__restrict__ b, float *__restrict__ Y) {
    __shared__ __attribute__((aligned(16))) float xs[kKChunk][kRows];
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < N; c0 += 256) {
        const int c = c0 + tid;
        float acc[kRows];
        const float bias = (c < N) ? b[c] : 0.0f;
#pragma unroll
        for (int r = 0; r < kRows; ++r) acc[r] = bias;
        for (int k0 = 0; k0 < K; k0 += kKChunk) {
            const int kc = min(kKChunk, K - k0);
            __syncthreads();
            for (int e = tid; e < kc * kRows; e += 256) {
                const int r = e / kc, k = e - r * kc;  // coalesced along k within a row
                const int64_t row = row0 + r;
                xs[k][r] = (row < m) ? X[row * K + k0 + k] : 0.0f;
            }
            __syncthreads();
            if (c < N) {
                for (int k = 0; k < kc; ++k) {
                    const float w = W[(int64_t)(k0 + k) * N + c];
                    const float4 x0 = *reinterpret_cast<const float4 *>(&xs[k][0]);
                    const float4 x1 = *reinterpret_cast<const float4 *>(&xs[k][4]);
                    const float4 x2 = *reinterpret_cast<const float4 *>(&xs[k][8]);
                    const float4 x3 = *reinterpret_cast<const float4 *>(&xs[k][12]);
                    acc[0] = fmaf(x0.x, w, acc[0]);   acc[1] = fmaf(x0.y, w, acc[1]);
                    acc[2] = fmaf(x0.z, w, acc[2]);   acc[3] = fmaf(x0.w, w, acc[3]);
                    acc[4] = fmaf(x1.x, w, acc[4]);   acc[5] = fmaf(x1.y, w, acc[5]);
                    acc[6] = fmaf(x1.z, w, acc[6]);   acc[7] = fmaf(x1.w, w, acc[7]);
                    acc[8] = fmaf(x2.x, w, acc[8]);   acc[9] = fmaf(x2.y, w, acc[9]);
                    acc[10] = fmaf(x2.z, w, acc[10]); acc[11] = fmaf(x2.w, w, acc[11]);
                    acc[12] = fmaf(x3.x, w, acc[12]); acc[13] = fmaf(x3.y, w, acc[13]);
                    acc[14] = fmaf(x3.z, w, acc[14]); acc[15] = fmaf(x3.w, w, acc[15]);
                }
            }
        }
        if (c < N) {
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int64_t row = row0 + r;
                if (row < m) Y[row * N + c] = RELU ? fmaxf(acc[r], 0.0f) : acc[r];  // feedforward_network.py:19
            }
        }
    }
}

// np.nan_to_num(np.divide(x - mean, std)) (dynamics_model.py:228-229): NaN -> 0, +-Inf -> +-max.
__device__ __forceinline__ float normalise_one(float x, float mean, float stdv) {
    const float v = (x - mean) / stdv;
    if (isnan(v)) return 0.0f;
    if (isinf(v)) return v > 0.0f ? FLT_MAX : -FLT_MAX;
    return v;
}

struct DynDims {
    int32_t state_dim, act_dim, H;
};

// S[0] = s0: one state per m / s0_rows consecutive rows (s0_rows == 1: np.tile, dynamics_model.py:215-217)
__global__ __launch_bounds__(256) void dyn_init_kernel(int64_t m, int d, const float *__restrict__ s0,
                                                       int64_t rows_per_state, float *__restrict__ S0) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * d) return;
    const int64_t row = e / d;
    const int k = (int)(e - row * d);
    S0[e] = s0[(row / rows_per_state) * d + k];   // rows_per_state = m / s0_rows
}

// x[m][d+a] = normalised (S_t, A[:, t])   (dynamics_model.py:228-230)
__global__ __launch_bounds__(256) void dyn_prepare_kernel(int64_t m, DynDims dd, int t, ssc_norm nm,
                                                          const float *__restrict__ St,
                                                          const float *__restrict__ A, float *__restrict__ x) {
    const int in = dd.state_dim + dd.act_dim;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * in) return;
    const int64_t row = e / in;
    const int k = (int)(e - row * in);
    float v;
    if (k < dd.state_dim)
        v = normalise_one(St[row * dd.state_dim + k], nm.mean_x[k], nm.std_x[k]);
    else {
        const int a = k - dd.state_dim;
        v = normalise_one(A[(row * dd.H + t) * dd.act_dim + a], nm.mean_y[a], nm.std_y[a]);
    }
    x[e] = v;
}

// S[t+1] = S[t] + z * std_z + mean_z   (dynamics_model.py:234-237)
__global__ __launch_bounds__(256) void dyn_update_kernel(int64_t m, int d, ssc_norm nm, const float *__restrict__ St,
                                                         const float *__restrict__ z, float *__restrict__ Sn) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * d) return;
    const int k = (int)(e % d);
    Sn[e] = St[e] + (z[e] * nm.std_z[k] + nm.mean_z[k]);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused fp32 forward simulation for the SMALL networks the reference ships (one hidden layer, depth <= 128, <= 4 inputs
// and outputs: the navigator's 1 x 32): ONE launch for all H steps instead of four launches per step (prepare, two
// layers, update) plus the action-sampling launch.  One row per lane, the state in registers for the whole horizon;
// the weights sit in LDS as one 8-float record per hidden unit {W1[0..3][j], b1[j], W2[j][0..2]} read as two broadcast
// ds_read_b128 (every lane reads the same address: conflict-free); per hidden unit <= 4 FMAs in, ReLU, <= 3 FMAs out.
// Arithmetic = the multi-launch path's: z-score by division (normalise_one), k-ordered fp32 FMA chains per unit,
// S' = S + (z std + mean).  VALU-bound (~8 ops per row, unit and step).  Candidate actions come from memory or are drawn
// in the kernel (the Philox stream of ssc_mpc_sample_actions); rows of problems masked out by d_problem_active are
// skipped per lane.
// ---------------------------------------------------------------------------------------------------------------
struct SmallSimArgs {
    int64_t m, rows_per_state;   // rows sharing one start state
    int32_t H, d, a, depth;
    const float *W1, *b1, *W2, *b2;   // [in][depth], [depth], [depth][d], [d]
    ssc_norm nm;
    const float *s0, *A;
    float *S, *A_out;
    int32_t sample, N;
    uint64_t seed, pid0, t;
    const uint64_t *t_base;
    float low[SSC_MAX_ACT], span[SSC_MAX_ACT];
    const uint8_t *active;
    const int32_t *live_list, *n_live;   // ssc_mpc_sampling: compact work list (threads -> rows of live problems)
};

constexpr int kSmallMaxDepth = 128;

__global__ __launch_bounds__(256) void dyn_small_sim_kernel(SmallSimArgs g) {
    __shared__ __attribute__((aligned(16))) float rec[kSmallMaxDepth][8];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int in = g.d + g.a;
    // compact work list: thread (slot * N + n) serves sample n of problem live_list[slot]; a block whose first slot lies
    // past the live count has nothing to do (block-uniform: in front of the barrier)
    int32_t n_live = 0;
    if (g.live_list != nullptr) {
        n_live = *g.n_live;
        if ((int64_t)blockIdx.x * 256 >= (int64_t)n_live * g.N) return;
    }
    for (int e = threadIdx.x; e < g.depth * 8; e += 256) {
        const int j = e >> 3, q = e & 7;
        float v = 0.0f;
        if (q < 4) v = (q < in) ? g.W1[q * g.depth + j] : 0.0f;
        else if (q == 4) v = g.b1[j];
        else v = (q - 5 < g.d) ? g.W2[j * g.d + (q - 5)] : 0.0f;
        rec[j][q] = v;
    }
    __syncthreads();
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool valid = gi < g.m;
    int64_t row = valid ? gi : g.m - 1;
    if (g.live_list != nullptr) {
        const int64_t slot = gi / g.N;
        valid = slot < n_live;
        row = valid ? (int64_t)g.live_list[slot] * g.N + (gi - slot * g.N) : 0;
    }
    if (!valid || (g.active != nullptr && g.active[row / g.N] == 0)) return;   // (no barrier below)
    float st[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < g.d) st[k] = g.s0[(row / g.rows_per_state) * g.d + k];
    const uint64_t sid = g.sample ? ((g.pid0 + (uint64_t)(row / g.N)) << 32) + (uint64_t)(row % g.N) : 0;
    const uint64_t tt = g.sample ? (g.t + (g.t_base != nullptr ? *g.t_base : 0)) * (uint64_t)((g.H * g.a + 3) / 4) : 0;
    u32x4 wcache = u32x4{0, 0, 0, 0};
    int wc = -1;
    for (int t = 0; t < g.H; ++t) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < g.d) g.S[((int64_t)t * g.m + row) * g.d + k] = st[k];          // dynamics_model.py:225
        float x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < g.d) x[k] = normalise_one(st[k], g.nm.mean_x[k], g.nm.std_x[k]);
#pragma unroll
        for (int ai = 0; ai < 3; ++ai)
            if (ai < g.a) {
                float av;
                if (g.sample) {
                    const int f = t * g.a + ai, c4 = f >> 2;
                    if (c4 != wc) {
                        wc = c4;
                        wcache = rng_words(g.seed, sid, tt + (uint64_t)c4, TAG_MPC);
                    }
                    av = uniform_f32(pick(wcache, (uint32_t)(f & 3)), g.low[ai], g.span[ai]);
                    if (g.A_out != nullptr) g.A_out[(row * g.H + t) * g.a + ai] = av;
                } else {
                    av = g.A[(row * g.H + t) * g.a + ai];
                }
                const float xa = normalise_one(av, g.nm.mean_y[ai], g.nm.std_y[ai]);
                // input g.d + ai (g.d + g.a <= 4)
                if (g.d + ai == 1) x[1] = xa; else if (g.d + ai == 2) x[2] = xa; else if (g.d + ai == 3) x[3] = xa;
            }
        float z[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < g.d) z[k] = g.b2[k];
#pragma unroll 4
        for (int j = 0; j < g.depth; ++j) {
            const f4 w = *reinterpret_cast<const f4 *>(&rec[j][0]);
            const f4 u = *reinterpret_cast<const f4 *>(&rec[j][4]);
            float h = u[0];
            h = fmaf(x[0], w[0], h);
            h = fmaf(x[1], w[1], h);
            h = fmaf(x[2], w[2], h);
            h = fmaf(x[3], w[3], h);
            h = fmaxf(h, 0.0f);                                                      // feedforward_network.py:19
            z[0] = fmaf(h, u[1], z[0]);
            z[1] = fmaf(h, u[2], z[1]);
            z[2] = fmaf(h, u[3], z[2]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < g.d) st[k] = st[k] + (z[k] * g.nm.std_z[k] + g.nm.mean_z[k]);    // :234-237
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < g.d) g.S[((int64_t)g.H * g.m + row) * g.d + k] = st[k];              // :240
}

// The same simulation for the shapes the shipped navigators have (state 2 or 3, ONE action, every byte offset below 2^31),
// compiled per shape and TWO ROWS PER LANE: the fp32 FMAs of a row pair are v_pk_fma_f32 (both halves against the same
// weight), so a hidden unit costs D + 1 packed FMAs in, two v_max, D packed FMAs out for two rows -- 7 instructions at
// D = 2 where the general kernel above issues 16.  SW: the weights are SGPR operands of those FMAs, fetched by scalar
// loads straight from W1 / b1 / W2 (four units a round; the loads of a wave hide under the FMAs of the others) -- with
// the LDS image of the general kernel the pair kernel spent 44 % of its wave cycles waiting on broadcast reads.
// The general kernel also spent ~45 % of its issue slots outside the network (64-bit divisions by runtime N, the generic
// d / a selects, a branch around every store); here the row -> (problem, sample) split is one 32-bit division per PAIR,
// the z-score's NaN / Inf mapping is two selects, and stores of switched-off rows are dropped by the buffer descriptor.
// Same arithmetic per row (k-ordered FMA chains, z-score by division), so both kernels give the same bits.
// normalise_one without branches: NaN -> 0, +-Inf -> +-FLT_MAX (v_med3_f32 clamps, a NaN is replaced first)
__device__ __forceinline__ float normalise_sel(float x, float mean, float stdv) {
    const float v = (x - mean) / stdv;
    return v != v ? 0.0f : __builtin_amdgcn_fmed3f(v, -FLT_MAX, FLT_MAX);
}

template <int D, bool SAMPLE, bool SW>
__global__ __launch_bounds__(256) void dyn_small_sim_pair_kernel(SmallSimArgs g) {
    __shared__ __attribute__((aligned(16))) float rec[kSmallMaxDepth][8];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u2v __attribute__((__vector_size__(2 * sizeof(unsigned int))));
    typedef const __attribute__((address_space(4))) float *cfloat;      // uniform reads of the weights: scalar loads
    constexpr int IN = D + 1;
    const uint32_t N = (uint32_t)g.N, m = (uint32_t)g.m;
    uint32_t total = m;
    if (g.live_list != nullptr) {
        total = (uint32_t)*g.n_live * N;
        if (blockIdx.x * 512u >= total) return;          // block-uniform, in front of the barrier
    }
    if (!SW) {
        for (int e = threadIdx.x; e < g.depth * 8; e += 256) {
            const int j = e >> 3, q = e & 7;
            float v = 0.0f;
            if (q < 4) v = (q < IN) ? g.W1[q * g.depth + j] : 0.0f;
            else if (q == 4) v = g.b1[j];
            else v = (q - 5 < D) ? g.W2[j * D + (q - 5)] : 0.0f;
            rec[j][q] = v;
        }
        __syncthreads();
    }
    const uint32_t pos = (blockIdx.x * 256u + threadIdx.x) * 2u;
    if (pos >= total) return;                             // (no barrier below)
    const bool two = pos + 1 < total;
    // position -> (problem slot, sample); the pair's second row is the next sample or the first one of the next slot
    uint32_t q0 = 0, n0 = pos, q1 = 0, n1 = pos + 1;
    if (SAMPLE) {
        q0 = pos / N; n0 = pos - q0 * N;
        q1 = q0; n1 = n0 + 1;
        if (n1 == N) { q1 = q0 + 1; n1 = 0; }
    }
    if (!two) { q1 = q0; n1 = n0; }
    uint32_t p0 = q0, p1 = q1;
    if (g.live_list != nullptr) { p0 = (uint32_t)g.live_list[q0]; p1 = (uint32_t)g.live_list[q1]; }
    const uint32_t r0 = SAMPLE ? p0 * N + n0 : n0, r1 = SAMPLE ? p1 * N + n1 : n1;
    // (the start states are requested with the mask bytes -- both need nothing but the problem index -- not behind them)
    const uint32_t rps = (uint32_t)g.rows_per_state;
    const uint32_t i0 = (SAMPLE && rps == N) ? p0 : r0 / rps, i1 = (SAMPLE && rps == N) ? p1 : r1 / rps;
    f2 st[D];
#pragma unroll
    for (int k = 0; k < D; ++k) st[k] = f2{g.s0[i0 * D + k], g.s0[i1 * D + k]};
    bool on0 = true, on1 = two;
    if (SAMPLE && g.active != nullptr) { on0 = g.active[p0] != 0; on1 = on1 && g.active[p1] != 0; }
    if (!on0 && !on1) return;
    // candidate actions: Philox words of (problem, sample), four (t, action) slots per call (ssc_mpc_sample_actions)
    const uint32_t c1_0 = (uint32_t)(g.pid0 + p0), c1_1 = (uint32_t)(g.pid0 + p1);
    const uint64_t tt = SAMPLE ? (g.t + (g.t_base != nullptr ? *g.t_base : 0)) * (uint64_t)((g.H + 3) / 4) : 0;
    u32x4 wa = u32x4{0, 0, 0, 0}, wb = wa;
    // The stores go through buffer descriptors: a row that is switched off (its problem inactive, or the odd row past the
    // end) gets an offset past the descriptor's range and the hardware drops the store -- no branch around any of them.
    const uint32_t s_bytes = (uint32_t)((size_t)(g.H + 1) * m * D * sizeof(float));
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(g.S, 0, (int)s_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(g.A_out, 0, g.A_out != nullptr ? (int)((size_t)m * g.H * sizeof(float)) : 0, 0x00020000);
    constexpr uint32_t kOff = 0x80000000u;               // past any range the launcher admits (< 2^31 bytes)
    uint32_t o0 = on0 ? r0 * (uint32_t)(D * sizeof(float)) : kOff, o1 = on1 ? r1 * (uint32_t)(D * sizeof(float)) : kOff;
    const uint32_t ao0 = on0 ? r0 * (uint32_t)g.H * 4u : kOff, ao1 = on1 ? r1 * (uint32_t)g.H * 4u : kOff;
    const uint32_t step_bytes = m * (uint32_t)(D * sizeof(float));
    auto put_state = [&]() {
        if (D == 2) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, f2{st[0][0], st[1][0]}), s_rsrc, (int)o0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, f2{st[0][1], st[1][1]}), s_rsrc, (int)o1, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) {       // (vector elements are not addressable: copy them out before the bit cast)
                const float v0 = st[k][0], v1 = st[k][1];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), s_rsrc, (int)o0 + 4 * k, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), s_rsrc, (int)o1 + 4 * k, 0, 0);
            }
        }
    };
    const cfloat cW1 = (cfloat)g.W1, cb1 = (cfloat)g.b1, cW2 = (cfloat)g.W2;
    for (int t = 0; t < g.H; ++t) {
        put_state();                                                               // dynamics_model.py:225
        if (on0) o0 += step_bytes;
        if (on1) o1 += step_bytes;
        f2 x[IN];
#pragma unroll
        for (int k = 0; k < D; ++k)
            x[k] = f2{normalise_sel(st[k][0], g.nm.mean_x[k], g.nm.std_x[k]), normalise_sel(st[k][1], g.nm.mean_x[k], g.nm.std_x[k])};
        f2 av;
        if (SAMPLE) {
            if ((t & 3) == 0) {
                const uint64_t ctr = tt + (uint64_t)(t >> 2);
                const uint32_t c2 = (uint32_t)ctr, c3 = (uint32_t)(((ctr >> 32) << 8) | TAG_MPC);
                wa = philox4x32_10(n0, c1_0, c2, c3, (uint32_t)g.seed, (uint32_t)(g.seed >> 32));
                wb = philox4x32_10(n1, c1_1, c2, c3, (uint32_t)g.seed, (uint32_t)(g.seed >> 32));
            }
            av = f2{uniform_f32(pick(wa, (uint32_t)(t & 3)), g.low[0], g.span[0]), uniform_f32(pick(wb, (uint32_t)(t & 3)), g.low[0], g.span[0])};
            const float av0 = av[0], av1 = av[1];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av0), a_rsrc, (int)(ao0 + 4u * (uint32_t)t), 0, 0);   // (a null A_out: range 0)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av1), a_rsrc, (int)(ao1 + 4u * (uint32_t)t), 0, 0);
        } else {
            av = f2{g.A[(size_t)r0 * g.H + t], g.A[(size_t)r1 * g.H + t]};
        }
        x[D] = f2{normalise_sel(av[0], g.nm.mean_y[0], g.nm.std_y[0]), normalise_sel(av[1], g.nm.mean_y[0], g.nm.std_y[0])};
        f2 z[D];
#pragma unroll
        for (int k = 0; k < D; ++k) z[k] = f2{g.b2[k], g.b2[k]};
        if (SW) {
            // four hidden units per round: W1[k][j .. j + 3], b1[j .. j + 3], W2[j .. j + 3][:] as scalar loads (SGPR operands of
            // the packed FMAs) -- the LDS stays out of the loop
            for (int j = 0; j < g.depth; j += 4) {            // (depth % 4 == 0: the launcher's condition for this variant)
                float w[IN][4], bb[4], u[4 * D];
#pragma unroll
                for (int k = 0; k < IN; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[k][e] = cW1[k * g.depth + j + e];
#pragma unroll
                for (int e = 0; e < 4; ++e) bb[e] = cb1[j + e];
#pragma unroll
                for (int e = 0; e < 4 * D; ++e) u[e] = cW2[j * D + e];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f2 h = f2{bb[e], bb[e]};
#pragma unroll
                    for (int k = 0; k < IN; ++k) h = __builtin_elementwise_fma(x[k], f2{w[k][e], w[k][e]}, h);
                    h = f2{fmaxf(h[0], 0.0f), fmaxf(h[1], 0.0f)};                     // feedforward_network.py:19
#pragma unroll
                    for (int k = 0; k < D; ++k) z[k] = __builtin_elementwise_fma(h, f2{u[e * D + k], u[e * D + k]}, z[k]);
                }
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < g.depth; ++j) {
                const f4 w = *reinterpret_cast<const f4 *>(&rec[j][0]);
                const f4 u = *reinterpret_cast<const f4 *>(&rec[j][4]);
                f2 h = f2{u[0], u[0]};
#pragma unroll
                for (int k = 0; k < IN; ++k) h = __builtin_elementwise_fma(x[k], f2{w[k], w[k]}, h);
                h = f2{fmaxf(h[0], 0.0f), fmaxf(h[1], 0.0f)};                         // feedforward_network.py:19
#pragma unroll
                for (int k = 0; k < D; ++k) z[k] = __builtin_elementwise_fma(h, f2{u[1 + k], u[1 + k]}, z[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {                                              // :234-237
            st[k][0] = st[k][0] + (z[k][0] * g.nm.std_z[k] + g.nm.mean_z[k]);
            st[k][1] = st[k][1] + (z[k][1] * g.nm.std_z[k] + g.nm.mean_z[k]);
        }
    }
    put_state();                                                                   // :240
}

// the shapes the fused fp32 kernel takes
static bool small_sim_shape(const ssc_mlp_desc *mlp, int state_dim, int act_dim) {
    return mlp->n_layers == 2 && mlp->dims[1] <= kSmallMaxDepth && state_dim <= 3 && state_dim + act_dim <= 4 && act_dim <= 3;
}

static int launch_small_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, const ssc_mpc_sampling *sp, int64_t m, int32_t H,
                            int state_dim, int act_dim, const float *d_s0, int64_t s0_rows, const float *d_A, float *d_A_out,
                            float *d_S, hipStream_t s) {
    SmallSimArgs g{};
    g.m = m; g.rows_per_state = m / s0_rows; g.H = H; g.d = state_dim; g.a = act_dim; g.depth = mlp->dims[1];
    g.W1 = mlp->W[0]; g.b1 = mlp->b[0]; g.W2 = mlp->W[1]; g.b2 = mlp->b[1];
    g.nm = *norm; g.s0 = d_s0; g.A = d_A; g.S = d_S; g.A_out = d_A_out;
    g.N = 1;
    if (sp != nullptr) {
        g.sample = 1; g.N = sp->n_samples; g.seed = sp->seed; g.pid0 = sp->problem_id0; g.t = sp->t; g.t_base = sp->d_t_base;
        for (int a = 0; a < SSC_MAX_ACT; ++a) {
            g.low[a] = a < act_dim ? sp->low[a] : 0.0f;
            g.span[a] = a < act_dim ? sp->high[a] - sp->low[a] : 0.0f;
        }
        g.active = sp->d_problem_active;
        if (sp->d_live_list != nullptr && sp->d_n_live != nullptr) { g.live_list = sp->d_live_list; g.n_live = sp->d_n_live; }
    }
    // the navigators' own shapes (state 2 or 3, one action) with every byte offset below 2^31: two rows per lane
    const bool pair = (state_dim == 2 || state_dim == 3) && act_dim == 1 && (int64_t)m * (H + 1) * state_dim * 4 < (1ll << 31) &&
                      (int64_t)m * H * 4 < (1ll << 31) &&
                      (sp == nullptr || (sp->n_samples >= 1 && sp->problem_id0 + (uint64_t)(m / sp->n_samples) < (1ull << 32)));
    if (pair) {
        const dim3 grid((unsigned)((m + 511) / 512));
        // weights as scalar loads, four hidden units a round, when the depth allows; else broadcast reads of an LDS image
        // (measured at 1 Mi rows x 4 steps, depth 32: 27 us of kernel at 84 % VALU busy against 33 us waiting on the LDS;
        // the general kernel: 50 us)
        const bool sw = g.depth % 4 == 0;
#define SSC_PAIR(DD, SS) do { if (sw) hipLaunchKernelGGL((dyn_small_sim_pair_kernel<DD, SS, true>), grid, dim3(256), 0, s, g); \
                              else hipLaunchKernelGGL((dyn_small_sim_pair_kernel<DD, SS, false>), grid, dim3(256), 0, s, g); } while (0)
        if (state_dim == 2) { if (sp != nullptr) SSC_PAIR(2, true); else SSC_PAIR(2, false); }
        else { if (sp != nullptr) SSC_PAIR(3, true); else SSC_PAIR(3, false); }
#undef SSC_PAIR
        return check_launch("dyn_small_sim_pair_kernel");
    }
    hipLaunchKernelGGL(dyn_small_sim_kernel, dim3(blocks_for(m)), dim3(256), 0, s, g);
    return check_launch("dyn_small_sim_kernel");
}

int validate_mlp(const ssc_mlp_desc *mlp, const char *who) {
    if (mlp == nullptr) return set_error(SSC_EINVAL, "%s: mlp NULL", who);
    if (mlp->n_layers < 1 || mlp->n_layers > SSC_MAX_LAYERS)
        return set_error(SSC_EINVAL, "%s: n_layers %d not in [1, %d]", who, mlp->n_layers, SSC_MAX_LAYERS);
    for (int l = 0; l <= mlp->n_layers; ++l)
        if (mlp->dims[l] < 1 || mlp->dims[l] > 65536) return set_error(SSC_EINVAL, "%s: bad dims[%d]", who, l);
    for (int l = 0; l < mlp->n_layers; ++l)
        if (!mlp->W[l] || !mlp->b[l]) return set_error(SSC_EINVAL, "%s: NULL weight pointer (layer %d)", who, l);
    return SSC_OK;
}

static int max_width(const ssc_mlp_desc *mlp) {
    int w = 0;
    for (int l = 0; l <= mlp->n_layers; ++l) w = mlp->dims[l] > w ? mlp->dims[l] : w;
    return w;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

void launch_mlp_layer_f32(bool relu, int64_t m, int K, int N, const float *X, const float *W, const float *b, float *Y,
                          hipStream_t s) {
    const unsigned grid = (unsigned)((m + kRows - 1) / kRows);
    if (relu) hipLaunchKernelGGL(mlp_layer_f32_kernel<true>, dim3(grid), dim3(256), 0, s, m, K, N, X, W, b, Y);
    else hipLaunchKernelGGL(mlp_layer_f32_kernel<false>, dim3(grid), dim3(256), 0, s, m, K, N, X, W, b, Y);
}

// fp32 path: y = net(x); act buffers a0/a1 of m*maxw floats each.
int mlp_forward_f32(const ssc_mlp_desc *mlp, int64_t m, const float *x, float *y, float *a0, float *a1,
                    hipStream_t s) {
    const float *cur = x;
    const unsigned grid = (unsigned)((m + kRows - 1) / kRows);
    for (int l = 0; l < mlp->n_layers; ++l) {
        const bool last = (l == mlp->n_layers - 1);
        float *dst = last ? y : ((l & 1) ? a1 : a0);
        if (last)
            hipLaunchKernelGGL(mlp_layer_f32_kernel<false>, dim3(grid), dim3(256), 0, s, m, mlp->dims[l],
                               mlp->dims[l + 1], cur, mlp->W[l], mlp->b[l], dst);
        else
            hipLaunchKernelGGL(mlp_layer_f32_kernel<true>, dim3(grid), dim3(256), 0, s, m, mlp->dims[l],
                               mlp->dims[l + 1], cur, mlp->W[l], mlp->b[l], dst);
        cur = dst;
    }
    return check_launch("mlp_forward_f32");
}

// bf16-MFMA path (dyn_mfma.hip)
bool dyn_mfma_supported(const ssc_mlp_desc *mlp, int state_dim, int act_dim);
size_t dyn_mfma_workspace_bytes(const ssc_mlp_desc *mlp);
int dyn_mfma_prepare(const ssc_mlp_desc *mlp, const ssc_norm *norm, void *ws, hipStream_t s);
int dyn_mfma_forward_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, int64_t m, int32_t H, int32_t state_dim,
                         int32_t act_dim, const float *d_s0, int64_t s0_rows, const float *d_A, float *d_S,
                         void *ws, bool prepared, hipStream_t s);
int dyn_mfma_mlp_forward(const ssc_mlp_desc *mlp, int64_t m, const float *d_x, float *d_y, void *ws, bool prepared,
                         hipStream_t s);
int dyn_mfma_forward_sim_sampled(const ssc_mlp_desc *mlp, const ssc_norm *norm, const ssc_mpc_sampling *sp, int64_t m,
                                 int32_t H, int32_t state_dim, int32_t act_dim, const float *d_s0, int64_t s0_rows,
                                 float *d_A_out, float *d_S, void *ws, bool prepared, hipStream_t s);

static bool is_mfma(int precision) { return precision == SSC_PREC_BF16_MFMA || precision == SSC_PREC_BF16_MFMA_PREPARED; }

}  // namespace ssc

using namespace ssc;

extern "C" {

size_t ssc_mlp_workspace_bytes(const ssc_mlp_desc *mlp, int64_t m, int precision) {
    if (mlp == nullptr || m < 0 || mlp->n_layers < 1 || mlp->n_layers > SSC_MAX_LAYERS) return 0;
    if (is_mfma(precision)) return dyn_mfma_workspace_bytes(mlp);
    return 2 * align256((size_t)m * max_width(mlp) * sizeof(float));
}

int ssc_mlp_forward(const ssc_mlp_desc *mlp, int64_t m, const float *d_x, float *d_y, int precision,
                    void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    if (int rc = validate_mlp(mlp, "ssc_mlp_forward")) return rc;
    SSC_REQUIRE(m >= 0, "ssc_mlp_forward: m < 0");
    if (m == 0) return SSC_OK;
    SSC_REQUIRE(d_x && d_y, "ssc_mlp_forward: NULL device pointer");
    const size_t need = ssc_mlp_workspace_bytes(mlp, m, precision);
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= need, "ssc_mlp_forward: workspace %zu < %zu bytes",
                workspace_bytes, need);
    if (is_mfma(precision)) {
        if (!dyn_mfma_supported(mlp, mlp->dims[0], 0))
            return set_error(SSC_EUNSUPPORTED, "ssc_mlp_forward: MFMA path needs 1-2 hidden layers of equal depth "
                                               "<= 512, in <= 12 (<= 10 with 2 layers deeper than 128), out <= 8");
        return dyn_mfma_mlp_forward(mlp, m, d_x, d_y, d_workspace, precision == SSC_PREC_BF16_MFMA_PREPARED,
                                    as_stream(stream));
    }
    SSC_REQUIRE(precision == SSC_PREC_F32, "ssc_mlp_forward: unknown precision %d", precision);
    float *a0 = static_cast<float *>(d_workspace);
    float *a1 = reinterpret_cast<float *>(static_cast<char *>(d_workspace) + need / 2);
    return mlp_forward_f32(mlp, m, d_x, d_y, a0, a1, as_stream(stream));
}

int ssc_dyn_prepare(const ssc_mlp_desc *mlp, const ssc_norm *norm, void *d_workspace, size_t workspace_bytes,
                    ssc_stream_t stream) {
    if (int rc = validate_mlp(mlp, "ssc_dyn_prepare")) return rc;
    if (!dyn_mfma_supported(mlp, mlp->dims[0], 0))
        return set_error(SSC_EUNSUPPORTED, "ssc_dyn_prepare: the MFMA path does not cover this network");
    const size_t need = dyn_mfma_workspace_bytes(mlp);
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= need, "ssc_dyn_prepare: workspace %zu < %zu bytes",
                workspace_bytes, need);
    return dyn_mfma_prepare(mlp, norm, d_workspace, as_stream(stream));
}

size_t ssc_dyn_workspace_bytes(const ssc_mlp_desc *mlp, int64_t m, int precision) {
    if (mlp == nullptr || m < 0 || mlp->n_layers < 1 || mlp->n_layers > SSC_MAX_LAYERS) return 0;
    if (is_mfma(precision)) return dyn_mfma_workspace_bytes(mlp);
    // x [m][in] + z [m][out] + two activation buffers
    return align256((size_t)m * mlp->dims[0] * 4) + align256((size_t)m * mlp->dims[mlp->n_layers] * 4) +
           2 * align256((size_t)m * max_width(mlp) * 4);
}

int ssc_dyn_forward_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, int64_t m, int32_t H, int32_t state_dim,
                        int32_t act_dim, const float *d_s0, int64_t s0_rows, const float *d_A, float *d_S,
                        int precision, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    if (int rc = validate_mlp(mlp, "ssc_dyn_forward_sim")) return rc;
    SSC_REQUIRE(norm != nullptr, "ssc_dyn_forward_sim: norm NULL");
    SSC_REQUIRE(m >= 0 && H >= 0, "ssc_dyn_forward_sim: m = %lld, H = %d", (long long)m, H);
    SSC_REQUIRE(state_dim >= 1 && state_dim <= SSC_MAX_STATE && act_dim >= 1 && act_dim <= SSC_MAX_ACT,
                "ssc_dyn_forward_sim: state_dim %d / act_dim %d out of range", state_dim, act_dim);
    SSC_REQUIRE(mlp->dims[0] == state_dim + act_dim && mlp->dims[mlp->n_layers] == state_dim,
                "ssc_dyn_forward_sim: network is %d -> %d, expected %d -> %d", mlp->dims[0],
                mlp->dims[mlp->n_layers], state_dim + act_dim, state_dim);
    SSC_REQUIRE(s0_rows >= 1 && (m == 0 || m % s0_rows == 0), "ssc_dyn_forward_sim: s0_rows must divide m");
    if (m == 0) return SSC_OK;
    SSC_REQUIRE(d_s0 && d_S && (H == 0 || d_A), "ssc_dyn_forward_sim: NULL device pointer");
    const size_t need = ssc_dyn_workspace_bytes(mlp, m, precision);
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= need,
                "ssc_dyn_forward_sim: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    if (is_mfma(precision)) {
        if (!dyn_mfma_supported(mlp, state_dim, act_dim))
            return set_error(SSC_EUNSUPPORTED, "ssc_dyn_forward_sim: MFMA path needs 1-2 hidden layers of equal "
                                               "depth <= 512 (inputs <= 10 when 2 layers deeper than 128)");
        return dyn_mfma_forward_sim(mlp, norm, m, H, state_dim, act_dim, d_s0, s0_rows, d_A, d_S, d_workspace,
                                    precision == SSC_PREC_BF16_MFMA_PREPARED, s);
    }
    SSC_REQUIRE(precision == SSC_PREC_F32, "ssc_dyn_forward_sim: unknown precision %d", precision);
    if (H >= 1 && small_sim_shape(mlp, state_dim, act_dim))    // the shipped navigator shape: one fused launch
        return launch_small_sim(mlp, norm, nullptr, m, H, state_dim, act_dim, d_s0, s0_rows, d_A, nullptr, d_S, s);
    char *w = static_cast<char *>(d_workspace);
    float *x = reinterpret_cast<float *>(w);
    w += align256((size_t)m * mlp->dims[0] * 4);
    float *z = reinterpret_cast<float *>(w);
    w += align256((size_t)m * state_dim * 4);
    float *a0 = reinterpret_cast<float *>(w);
    w += align256((size_t)m * max_width(mlp) * 4);
    float *a1 = reinterpret_cast<float *>(w);
    const DynDims dd{state_dim, act_dim, H};
    const int in = state_dim + act_dim;
    hipLaunchKernelGGL(dyn_init_kernel, dim3(blocks_for(m * state_dim)), dim3(256), 0, s, m, state_dim, d_s0,
                       m / s0_rows, d_S);
    for (int t = 0; t < H; ++t) {
        const float *St = d_S + (size_t)t * m * state_dim;
        float *Sn = d_S + (size_t)(t + 1) * m * state_dim;
        hipLaunchKernelGGL(dyn_prepare_kernel, dim3(blocks_for(m * in)), dim3(256), 0, s, m, dd, t, *norm, St, d_A, x);
        if (int rc = mlp_forward_f32(mlp, m, x, z, a0, a1, s)) return rc;
        hipLaunchKernelGGL(dyn_update_kernel, dim3(blocks_for(m * state_dim)), dim3(256), 0, s, m, state_dim, *norm,
                           St, z, Sn);
    }
    return check_launch("ssc_dyn_forward_sim");
}

int ssc_mpc_forward_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, const ssc_mpc_sampling *sp, int64_t m, int32_t H,
                        int32_t state_dim, int32_t act_dim, const float *d_s0, int64_t s0_rows, float *d_A_out,
                        float *d_S, int precision, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    if (int rc = validate_mlp(mlp, "ssc_mpc_forward_sim")) return rc;
    SSC_REQUIRE(norm != nullptr && sp != nullptr, "ssc_mpc_forward_sim: NULL descriptor");
    SSC_REQUIRE(m >= 0 && H >= 1, "ssc_mpc_forward_sim: m = %lld, H = %d", (long long)m, H);
    SSC_REQUIRE(state_dim >= 1 && state_dim <= SSC_MAX_STATE && act_dim >= 1 && act_dim <= SSC_MAX_ACT,
                "ssc_mpc_forward_sim: state_dim %d / act_dim %d out of range", state_dim, act_dim);
    SSC_REQUIRE(mlp->dims[0] == state_dim + act_dim && mlp->dims[mlp->n_layers] == state_dim,
                "ssc_mpc_forward_sim: network is %d -> %d, expected %d -> %d", mlp->dims[0],
                mlp->dims[mlp->n_layers], state_dim + act_dim, state_dim);
    SSC_REQUIRE(sp->n_samples >= 1 && m % sp->n_samples == 0, "ssc_mpc_forward_sim: n_samples %d must divide m = %lld",
                sp->n_samples, (long long)m);
    SSC_REQUIRE(s0_rows >= 1 && (m == 0 || m % s0_rows == 0), "ssc_mpc_forward_sim: s0_rows must divide m");
    for (int a = 0; a < act_dim; ++a)
        SSC_REQUIRE(sp->low[a] <= sp->high[a], "ssc_mpc_forward_sim: low > high for action component %d", a);
    if (m == 0) return SSC_OK;
    SSC_REQUIRE(d_s0 && d_S, "ssc_mpc_forward_sim: NULL device pointer");
    const size_t need = ssc_dyn_workspace_bytes(mlp, m, precision);
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= need,
                "ssc_mpc_forward_sim: workspace %zu < %zu bytes", workspace_bytes, need);
    if (is_mfma(precision)) {
        if (!dyn_mfma_supported(mlp, state_dim, act_dim))
            return set_error(SSC_EUNSUPPORTED, "ssc_mpc_forward_sim: MFMA path needs 1-2 hidden layers of equal "
                                               "depth <= 512 (inputs <= 10 when 2 layers deeper than 128)");
        return dyn_mfma_forward_sim_sampled(mlp, norm, sp, m, H, state_dim, act_dim, d_s0, s0_rows, d_A_out, d_S,
                                            d_workspace, precision == SSC_PREC_BF16_MFMA_PREPARED, as_stream(stream));
    }
    SSC_REQUIRE(precision == SSC_PREC_F32, "ssc_mpc_forward_sim: unknown precision %d", precision);
    if (small_sim_shape(mlp, state_dim, act_dim))    // one fused launch, candidates drawn in the kernel (d_A_out optional)
        return launch_small_sim(mlp, norm, sp, m, H, state_dim, act_dim, d_s0, s0_rows, nullptr, d_A_out, d_S, as_stream(stream));
    // other fp32 shapes: the two-launch equivalent
    SSC_REQUIRE(d_A_out != nullptr, "ssc_mpc_forward_sim: the fp32 path materialises the action matrix: d_A_out NULL");
    if (int rc = ssc_mpc_sample_actions((int32_t)(m / sp->n_samples), sp->n_samples, H, act_dim, sp->low, sp->high, sp->seed,
                                        sp->problem_id0, sp->t, sp->d_t_base, d_A_out, stream))
        return rc;
    return ssc_dyn_forward_sim(mlp, norm, m, H, state_dim, act_dim, d_s0, s0_rows, d_A_out, d_S, precision, d_workspace,
                               workspace_bytes, stream);
}

}  // extern "C"
