// ddpg_train_wide.hip -- the DDPG learner step (DDPG_editted.train + update_target_net,
// DDPG_Baselines_editted/ddpg_editted.py:287-339) for ANY layer sizes and batch sizes, spread over the chip.
//
// The single-workgroup kernels (ddpg_train_fixed.hip: the shipped 64-32 shape; ddpg_train.hip: the step
// interpreter) keep the batch's activations and all four parameter vectors in ONE CU's LDS, which caps them at
// batch 64 and hidden layers <= 64.  The reference's own experiment grid goes to actor / critic 128-64 and 200-100
// (data/ddpg_baselines_summaries/hidden_layer_size_experiment/, ctor kwargs DDPG_Baselines_agent.py:86-92), and a
// vectorised actor-learner loop wants batches of hundreds to thousands.  Here:
//
//   * ddpg_wide_grad_kernel: the batch is tiled over workgroups, 16 rows each (the N of v_mfma_f32_16x16x4_f32).
//     A workgroup runs the whole forward / backward chain of its rows -- target networks, critic on (s, a), actor,
//     critic on (s, pi(s)), both backward passes -- with the rows' activations in LDS ([unit][16 rows]) and the
//     weights streamed from L2 as the MFMA A operand (they are read-only during the launch; 4 nets x <= 84 KB stay
//     L2-resident).  Every contraction, large or small, goes through one tile routine (exact fp32 products and
//     sums, so the fp64-oracle tolerance of the single-workgroup kernels carries over); a level's independent
//     contractions are flattened into (contraction, tile-pair) jobs over the 8 waves.  The workgroup's share of
//     every gradient (X^T dZ over its 16 rows: 4 MFMAs per 16 x 16 tile) goes to a per-workgroup slice in HBM.
//   * ddpg_wide_apply_kernel: one thread per parameter sums the per-workgroup partials IN WORKGROUP ORDER (bitwise
//     reproducible, no float atomics), then MpiAdam.update and the soft target update on that element.
//
// Two launches per training iteration (the iterations are a serial chain through the parameters); the Adam step
// counters are read-only during the launch sequence (iteration index added on the fly) and advanced by one
// one-thread launch at the end.
#include "ddpg_device.h"
#include "ssc_host.h"

namespace ssc {

constexpr int kWR = 16;           // batch rows per workgroup
constexpr int kWThreads = 512;    // 8 waves: two per SIMD, so one wave's L2 operand loads hide under the other's MFMAs
constexpr int kWWaves = kWThreads / 64;
constexpr int kMaxGemm = 24, kMaxLevel = 12, kMaxWg = 12, kMaxLn = 16, kLnPerLevel = 4;
constexpr int kLnParts = kWThreads / kWR;   // threads per batch row in the LayerNorm passes

enum : int { EPI_NONE = 0, EPI_RELU, EPI_TANH, EPI_MASK_RELU, EPI_MASK_TANH };

// out[m][row] = EPI( sum_kk A(m, kk) * B[kk][row] + bias[m] + add[m][row] ),  A(m, kk) = W[m * sm + kk * sk]
struct WGemm {
    const float *W;      // global: the h1 x h2 blocks (read-only during the launch); nullptr: the A operand is in LDS
    int32_t w_off;       // LDS float offset of the A operand inside the small-parameter image (W == nullptr)
    int32_t bias_off;    // LDS float offset of bias[M] inside the image, < 0: none
    int32_t sm, sk, M, K;
    int32_t b_off, out_off, add_off, aux_off;   // LDS float offsets; add_off / aux_off < 0: none
    int32_t epi;
};

// gradient of one layer: dW[k][u] = sum_rows X[k][row] * dZ[u][row], db[u] = sum_rows dZ[u][row]
struct WGrad {
    int32_t x_off, dz_off, in, out;   // LDS float offsets of the X rows [in][16] and the delta rows [out][16]
    int32_t gW, gb;                   // offsets into the flat gradient vector (actor first, then critic)
};

// LayerNorm (models_editted.py:45-46, 50-51, 85-86, 91-92) over the M units of a [M][16 rows] block, applied after the
// contractions of a level:
//   forward   x <- act((x - mean) * rstd * gamma + beta) in place; x-hat and 1/sigma are kept for the backward pass of the
//             trained networks (xhat_off / rstd_off >= 0);
//   backward  x holds dL/d(LayerNorm output) (the activation's derivative already applied by the contraction's epilogue) and
//             becomes dL/dz = rstd * (dxh - mean_u(dxh) - xhat * mean_u(dxh * xhat)), dxh = x * gamma; gamma / beta gradients
//             (sums over the workgroup's 16 rows) go to the gradient slice when gG >= 0.
struct WLn {
    int32_t bwd, x_off, xhat_off, rstd_off, g_off, b_off, M, act, gG, gB;
};

// The per-level tables.  They reach the kernel as arguments, but a scalar load from the kernel-argument segment that
// misses the (cold) scalar cache costs ~500 cycles and the job dispatch walks these tables level by level -- two to
// three dependent misses per level, ten levels: more than the arithmetic of a 64-32 network.  Every workgroup
// therefore copies the tables into LDS with ONE vector load per thread at kernel start and reads them from there
// (lds_uniform: broadcast read + v_readfirstlane, so that dispatch and addressing still run on the scalar unit).
struct WideTables {
    int32_t level_first[kMaxLevel + 1];
    int32_t ln_first[kMaxLevel + 1];   // LayerNorm ops that follow the contractions of a level
    int32_t n_seg, n_big, off_scr;     // off_scr: LDS scratch of the LayerNorm reductions, [kLnPerLevel][2][kLnParts][16]
    WLn ln[kMaxLn];
    WGemm gemm[kMaxGemm];
    WGrad wg[kMaxWg];
    // Everything of the four parameter vectors except their h1 x h2 blocks (first layers, biases, output layers, the
    // critics' action rows: a few KB) is copied into LDS at kernel start, segment by segment; the h1 x h2 blocks are
    // touched once (one load per 128-byte line) so that the contractions find them in this XCD's L2.
    struct { const float *src; int32_t dst, n; } seg[8];
    struct { const float *src; int32_t n, pad; } big[4];
};
static_assert(sizeof(WideTables) % 4 == 0 && sizeof(WideTables) <= 2 * kWThreads * 4, "two dwords per thread copy the tables");

struct WideArgs {
    ssc_replay_view rp;
    const int32_t *batch_idx;    // [batch] of this iteration
    int32_t batch, obs_dim, act_dim;
    float gamma, obs_clip;
    int32_t off_S, off_S2, off_ACT, off_TACT, off_RT;   // LDS float offsets; RT: rows r, t, y, q, q', qpi, dq, dqb
    int32_t n_gemm, n_level, td_level, n_wg;
    float *gpart;   // [n_blocks][n_params]
    float *lpart;   // [n_blocks][2]
    int32_t n_params;
    int32_t off_tab;             // LDS float offset of the copy of `tab`
    WideTables tab;
};

template <class T> __device__ __forceinline__ T lds_uniform(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized tables");
    union { T t; uint32_t w[sizeof(T) / 4]; } u;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) u.w[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return u.t;
}

enum : int { RT_R = 0, RT_T, RT_Y, RT_Q, RT_QT, RT_QPI, RT_DQ, RT_DQB, RT_ROWS };

// The hot loops below are written branch-free: an operand outside the valid range is read from a clamped (valid)
// address and zeroed with a select -- a predicated load costs hipcc a basic block and a full s_waitcnt per load, which
// serialises what should be sixteen loads in flight.  Offsets are 32-bit (SGPR base + VGPR offset addressing).
__device__ __forceinline__ float act_epi(int epi, float v, float aux) {
    if (epi == EPI_RELU) return fmaxf(v, 0.0f);
    if (epi == EPI_TANH) return tanh_fast(v);
    if (epi == EPI_MASK_RELU) return aux > 0.0f ? v : 0.0f;
    if (epi == EPI_MASK_TANH) return v * (1.0f - aux * aux);
    return v;
}

__device__ __forceinline__ void gemm_epilogue(const WGemm &g, float *lds, int tile0, int lane, const f32x4m &acc0, const f32x4m &acc1) {
    const int lm = lane & 15, lk = lane >> 4;
    // all (up to) 24 LDS reads of the epilogue are issued unconditionally -- an absent bias / addend / derivative row
    // reads the output row instead and is dropped by a select -- so that they share ONE wait (a conditional read costs
    // a basic block and a full lgkmcnt(0) each: 24 serialised LDS round trips were most of a small level's time)
    const bool has_bias = g.bias_off >= 0, has_add = g.add_off >= 0, has_aux = g.aux_off >= 0;
    const int bias_o = has_bias ? g.bias_off : g.out_off, add_o = has_add ? g.add_off : g.out_off, aux_o = has_aux ? g.aux_off : g.out_off;
    float bias[8], add[8], aux[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int mc = min((tile0 + (e >> 2)) * 16 + 4 * lk + (e & 3), g.M - 1);
        bias[e] = lds[bias_o + mc];
        add[e] = lds[add_o + mc * kWR + lm];
        aux[e] = lds[aux_o + mc * kWR + lm];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int m = (tile0 + (e >> 2)) * 16 + 4 * lk + (e & 3);
        float x = (e < 4 ? acc0[e & 3] : acc1[e & 3]) + (has_bias ? bias[e] : 0.0f) + (has_add ? add[e] : 0.0f);
        const float r = act_epi(g.epi, x, has_aux ? aux[e] : 0.0f);
        if (m < g.M) lds[g.out_off + m * kWR + lm] = r;
    }
}

// One job of a contraction: up to two 16-unit tiles (they share the B operand) for the workgroup's 16 rows.
// A operand in the LDS image (first / output layers, action rows -- short contractions); reads batched 8 k-steps at
// a time (one wait per 24 reads).
__device__ __forceinline__ void gemm_job_lds(const WGemm &g, float *lds, int tile0, int lane) {
    constexpr int KU = 8;
    const int lm = lane & 15, lk = lane >> 4;
    const int m0 = tile0 * 16 + lm, m1 = m0 + 16;
    const bool v0 = m0 < g.M, v1 = m1 < g.M;
    const float *w0 = lds + g.w_off + min(m0, g.M - 1) * g.sm;
    const float *w1 = lds + g.w_off + min(m1, g.M - 1) * g.sm;
    const float *b = lds + g.b_off + lm;
    f32x4m acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int steps = (g.K + 3) >> 2;
    for (int s0 = 0; s0 < steps; s0 += KU) {
        float a0[KU], a1[KU], bv[KU];
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            const int kc = min(4 * (s0 + j) + lk, g.K - 1);
            a0[j] = w0[kc * g.sk];
            a1[j] = w1[kc * g.sk];
            bv[j] = b[kc * kWR];
        }
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            const bool ok = 4 * (s0 + j) + lk < g.K;
            acc0 = mfma4((ok && v0) ? a0[j] : 0.0f, ok ? bv[j] : 0.0f, acc0);
            acc1 = mfma4((ok && v1) ? a1[j] : 0.0f, ok ? bv[j] : 0.0f, acc1);
        }
    }
    gemm_epilogue(g, lds, tile0, lane, acc0, acc1);
}

// A operand streamed from L2 (the h1 x h2 blocks) in chunks of KU k-steps with one chunk of look-ahead, so that a
// chunk's round trip runs under the previous chunk's 2 KU MFMAs.
__device__ __forceinline__ void gemm_job_l2(const WGemm &g, float *lds, int tile0, int lane) {
    constexpr int KU = 16;
    const int lm = lane & 15, lk = lane >> 4;
    const int m0 = tile0 * 16 + lm, m1 = m0 + 16;
    const bool v0 = m0 < g.M, v1 = m1 < g.M;
    const float *__restrict__ W = g.W;
    const uint32_t o0 = (uint32_t)(min(m0, g.M - 1) * g.sm), o1 = (uint32_t)(min(m1, g.M - 1) * g.sm);
    const float *b = lds + g.b_off + lm;
    f32x4m acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int steps = (g.K + 3) >> 2;
    float a0[KU], a1[KU], n0[KU], n1[KU];
    auto fetch = [&](int s0, float (&x0)[KU], float (&x1)[KU]) {
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            const uint32_t ko = (uint32_t)(min(4 * (s0 + j) + lk, g.K - 1) * g.sk);
            x0[j] = W[o0 + ko];
            x1[j] = W[o1 + ko];
        }
    };
    fetch(0, a0, a1);
    for (int s0 = 0; s0 < steps; s0 += KU) {
        fetch(min(s0 + KU, steps - 1), n0, n1);      // in flight under this chunk's MFMAs (the last one re-reads valid data)
        float bv[KU];
#pragma unroll
        for (int j = 0; j < KU; ++j) bv[j] = b[min(4 * (s0 + j) + lk, g.K - 1) * kWR];
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            const bool ok = 4 * (s0 + j) + lk < g.K;
            acc0 = mfma4((ok && v0) ? a0[j] : 0.0f, ok ? bv[j] : 0.0f, acc0);
            acc1 = mfma4((ok && v1) ? a1[j] : 0.0f, ok ? bv[j] : 0.0f, acc1);
        }
#pragma unroll
        for (int j = 0; j < KU; ++j) { a0[j] = n0[j]; a1[j] = n1[j]; }
    }
    gemm_epilogue(g, lds, tile0, lane, acc0, acc1);
}

// One 16 x 16 tile of a layer's weight gradient (tile i over the inputs, tile j over the units): the 16 rows are the
// K of four MFMAs; lane group lk contracts rows 4 lk .. 4 lk + 3 (any pairing of k values is a valid contraction),
// so both operands come in with one 16-byte LDS read per lane.
__device__ __forceinline__ void wgrad_tile(const WGrad &w, const float *lds, float *__restrict__ gout, int ti, int tj, int lane) {
    const int lm = lane & 15, lk = lane >> 4;
    const int k = ti * 16 + lm, u = tj * 16 + lm;
    f4 xa = *reinterpret_cast<const f4 *>(lds + w.x_off + min(k, w.in - 1) * kWR + 4 * lk);
    f4 dz = *reinterpret_cast<const f4 *>(lds + w.dz_off + min(u, w.out - 1) * kWR + 4 * lk);
    if (k >= w.in) xa = f4{0.0f, 0.0f, 0.0f, 0.0f};
    if (u >= w.out) dz = f4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4m acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma4(xa[s], dz[s], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = ti * 16 + 4 * lk + i;
        if (kk < w.in && u < w.out) gout[(uint32_t)(w.gW + kk * w.out + u)] = acc[i];
    }
}

// The LayerNorm ops behind one level (<= kLnPerLevel of them, all between the same barriers).  Thread t serves batch row
// t & 15 and the units u = part, part + 32, ... (part = t >> 4); the per-row reductions go through LDS partials summed by
// every thread in part order, so the result does not depend on which wave did what.
__device__ __forceinline__ void ln_level(const WideTables *T, int first, int end, float *lds, float *__restrict__ gout, int tid) {
    const int row = tid & (kWR - 1), part = tid >> 4;
    float *scr = lds + T->off_scr;
    auto post = [&](int slot, float a, float b) {   // this thread's partial sums (the barriers around it are the caller's)
        scr[((slot * 2 + 0) * kLnParts + part) * kWR + row] = a;
        scr[((slot * 2 + 1) * kLnParts + part) * kWR + row] = b;
    };
    auto total = [&](int slot, int which) {
        float s = 0.0f;
#pragma unroll 8
        for (int q = 0; q < kLnParts; ++q) s += scr[((slot * 2 + which) * kLnParts + q) * kWR + row];
        return s;
    };
    float mean[kLnPerLevel], rstd[kLnPerLevel];
    // pass 1: forward -> sum of x; backward -> sums of dxh and dxh * xhat
    for (int li = first; li < end; ++li) {
        const WLn L = lds_uniform(&T->ln[li]);
        float s0 = 0.0f, s1 = 0.0f;
        for (int u = part; u < L.M; u += kLnParts) {
            const float x = lds[L.x_off + u * kWR + row];
            if (L.bwd) {
                const float dxh = x * lds[L.g_off + u];
                s0 += dxh;
                s1 = fmaf(dxh, lds[L.xhat_off + u * kWR + row], s1);
            } else {
                s0 += x;
            }
        }
        post(li - first, s0, s1);
    }
    __syncthreads();
    for (int li = first; li < end; ++li) {
        const int M = __builtin_amdgcn_readfirstlane(T->ln[li].M);
        mean[li - first] = total(li - first, 0) / (float)M;
        rstd[li - first] = total(li - first, 1) / (float)M;      // backward: mean_u(dxh * xhat)
    }
    __syncthreads();
    // pass 2 (forward only): sum of (x - mean)^2
    bool any_fwd = false;
    for (int li = first; li < end; ++li) {
        const WLn L = lds_uniform(&T->ln[li]);
        if (L.bwd) continue;
        any_fwd = true;
        float v = 0.0f;
        for (int u = part; u < L.M; u += kLnParts) {
            const float dlt = lds[L.x_off + u * kWR + row] - mean[li - first];
            v = fmaf(dlt, dlt, v);
        }
        post(li - first, v, 0.0f);
    }
    if (any_fwd) {   // block-uniform
        __syncthreads();
        for (int li = first; li < end; ++li) {
            const WLn L = lds_uniform(&T->ln[li]);
            if (!L.bwd) rstd[li - first] = 1.0f / sqrtf(total(li - first, 0) / (float)L.M + 1e-12f);   // tc.layers.layer_norm's epsilon
        }
    }
    // pass 3: the element-wise result
    for (int li = first; li < end; ++li) {
        const WLn L = lds_uniform(&T->ln[li]);
        if (L.bwd) {
            const float r = lds[L.rstd_off + row], m0 = mean[li - first], m1 = rstd[li - first];
            // gamma / beta gradients first (they read the incoming deltas): one thread per unit, the 16 rows in order
            if (L.gG >= 0) {
                for (int u = tid; u < L.M; u += kWThreads) {
                    float sg = 0.0f, sb = 0.0f;
#pragma unroll
                    for (int q = 0; q < kWR; ++q) {
                        const float dn = lds[L.x_off + u * kWR + q];
                        sg = fmaf(dn, lds[L.xhat_off + u * kWR + q], sg);
                        sb += dn;
                    }
                    gout[L.gG + u] = sg;
                    gout[L.gB + u] = sb;
                }
                __syncthreads();      // (block-uniform: L is)
            }
            for (int u = part; u < L.M; u += kLnParts) {
                const float xh = lds[L.xhat_off + u * kWR + row];
                const float dxh = lds[L.x_off + u * kWR + row] * lds[L.g_off + u];
                lds[L.x_off + u * kWR + row] = r * (dxh - m0 - xh * m1);
            }
        } else {
            const float m = mean[li - first], r = rstd[li - first];
            if (L.rstd_off >= 0 && part == 0) lds[L.rstd_off + row] = r;
            for (int u = part; u < L.M; u += kLnParts) {
                const float xh = (lds[L.x_off + u * kWR + row] - m) * r;
                if (L.xhat_off >= 0) lds[L.xhat_off + u * kWR + row] = xh;
                const float n = fmaf(xh, lds[L.g_off + u], lds[L.b_off + u]);
                lds[L.x_off + u * kWR + row] = L.act == EPI_TANH ? tanh_fast(n) : fmaxf(n, 0.0f);
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kWThreads) void ddpg_wide_grad_kernel(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef SSC_WIDE_DIAG
    const uint64_t t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: job dispatch runs on the scalar unit
    const int row0 = blockIdx.x * kWR;
    // ---- the tables: one dword per thread from the kernel-argument segment into LDS ----
#if defined(__HIP_DEVICE_COMPILE__)
    {
        typedef const __attribute__((address_space(4))) char *karg_bytes;
        typedef const __attribute__((address_space(4))) uint32_t *karg_words;
        const karg_words src = (karg_words)((karg_bytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WideArgs, tab));
#pragma unroll
        for (int e = tid; e < (int)(sizeof(WideTables) / 4); e += kWThreads) reinterpret_cast<uint32_t *>(lds + a.off_tab)[e] = src[e];
    }
#endif
    __syncthreads();
    const WideTables *T = reinterpret_cast<const WideTables *>(lds + a.off_tab);
    // ---- everything the launch needs from memory is requested up front, in one round trip where it can be:
    //   * the batch indices (the rows need a second, dependent one);
    //   * the small parameters -> LDS image (8 segments, up to kImgQ x 512 floats each per pass);
    //   * one load per 128-byte line of the h1 x h2 blocks (L2 warm-up: the previous launch rewrote every parameter,
    //     so the first touch of a line in this XCD goes to the memory side).
    const bool row_thread = tid < kWR;
    const bool valid = row0 + tid < a.batch;
    int64_t rec = 0;
    if (row_thread) rec = a.batch_idx[valid ? row0 + tid : a.batch - 1];
    constexpr int kImgQ = 2;
    {
        const float *seg_src[8];
        int seg_dst[8], seg_n[8], max_seg = 0;
#pragma unroll
        for (int si = 0; si < 8; ++si) {
            const auto sg = lds_uniform(&T->seg[si]);
            seg_src[si] = sg.src; seg_dst[si] = sg.dst; seg_n[si] = sg.n;     // (unused slots: n = 0, src = a valid pointer)
            max_seg = max(max_seg, sg.n);
        }
        for (int q0 = 0; q0 * kWThreads < max_seg; q0 += kImgQ) {
            float v[8][kImgQ];
#pragma unroll
            for (int si = 0; si < 8; ++si)
#pragma unroll
                for (int q = 0; q < kImgQ; ++q) v[si][q] = seg_src[si][max(min(tid + (q0 + q) * kWThreads, seg_n[si] - 1), 0)];
#pragma unroll
            for (int si = 0; si < 8; ++si)
#pragma unroll
                for (int q = 0; q < kImgQ; ++q) {
                    const int e = tid + (q0 + q) * kWThreads;
                    if (e < seg_n[si]) lds[seg_dst[si] + e] = v[si][q];
                }
        }
        float sink = 0.0f;
        const float *big_src[4];
        int big_n[4], max_big = 0;
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
            const auto bg = lds_uniform(&T->big[bi]);
            big_src[bi] = bg.src; big_n[bi] = bg.n;
            max_big = max(max_big, bg.n);
        }
        for (int e0 = 0; e0 < max_big; e0 += kWThreads * 32 * 2) {
            float v[4][2];
#pragma unroll
            for (int bi = 0; bi < 4; ++bi)
#pragma unroll
                for (int q = 0; q < 2; ++q) v[bi][q] = big_src[bi][max(min(e0 + (tid + q * kWThreads) * 32, big_n[bi] - 1), 0)];
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) sink += v[bi][0] + v[bi][1];
        }
        if (sink == 1.2345e-30f) lds[a.off_RT + RT_Y * kWR] = sink;   // keeps the loads; the row is rewritten below
    }
    // ---- ReplayBuffer.sample_batch rows of this workgroup (replay_buffer.py:79-91); rows past the batch shadow its
    // last record and carry zero weight in every loss ----
    if (row_thread) {
        const int r = tid;
        for (int k = 0; k < a.obs_dim; ++k) {   // obs0 / obs1 enter every network clipped (ddpg_editted.py:106-109)
            float s = a.rp.s[rec * a.obs_dim + k], s2 = a.rp.s2[rec * a.obs_dim + k];
            if (a.obs_clip > 0.0f) {
                s = fminf(fmaxf(s, -a.obs_clip), a.obs_clip);
                s2 = fminf(fmaxf(s2, -a.obs_clip), a.obs_clip);
            }
            lds[a.off_S + k * kWR + r] = s;
            lds[a.off_S2 + k * kWR + r] = s2;
        }
        for (int k = 0; k < a.act_dim; ++k) lds[a.off_ACT + k * kWR + r] = a.rp.a[rec * a.act_dim + k];
        lds[a.off_RT + RT_R * kWR + r] = a.rp.r[rec];
        lds[a.off_RT + RT_T * kWR + r] = a.rp.t[rec] ? 1.0f : 0.0f;
    }
    __syncthreads();
#ifdef SSC_WIDE_DIAG
    // diagnostic build (tools/exp_ddpg_wide_phases.py): cycle stamps of block 0 at every level boundary, written behind
    // the loss partials
    uint64_t stamp[kMaxLevel + 4];
    int n_stamp = 0;
    stamp[n_stamp++] = t_entry;
    stamp[n_stamp++] = __builtin_amdgcn_s_memtime();
#endif
    float *gout = a.gpart + (int64_t)blockIdx.x * a.n_params;
#ifdef SSC_WIDE_DIAG
    // the level sequence runs TWICE in the diagnostic build and the stamps are those of the second pass: same work,
    // but with the instruction cache warm -- the difference to the first pass is what a fresh launch pays for code fetch
    for (int rep = 0; rep < SSC_WIDE_DIAG; ++rep) {
    n_stamp = 2;
    stamp[1] = __builtin_amdgcn_s_memtime();
#endif
    for (int lv = 0; lv < a.n_level; ++lv) {
        // the level's independent contractions as (contraction, tile-pair) jobs, dealt round-robin to the waves
        int job = wave;
        const int g_first = __builtin_amdgcn_readfirstlane(T->level_first[lv]), g_end = __builtin_amdgcn_readfirstlane(T->level_first[lv + 1]);
        for (int gi = g_first; gi < g_end; ++gi) {
            const int M = __builtin_amdgcn_readfirstlane(T->gemm[gi].M);
            const int pairs = (((M + 15) >> 4) + 1) >> 1;
            if (job < pairs) {
                const WGemm g = lds_uniform(&T->gemm[gi]);
                if (g.W == nullptr) {
                    for (; job < pairs; job += kWWaves) gemm_job_lds(g, lds, 2 * job, lane);
                } else {
                    for (; job < pairs; job += kWWaves) gemm_job_l2(g, lds, 2 * job, lane);
                }
            }
            job -= pairs;
        }
        __syncthreads();
        if (lv == a.td_level) {
            // target_Q = r + (1 - terminal) * gamma * Q'(s2, pi'(s2))  (:132-133); critic loss mean((Q - y)^2) (:181),
            // actor loss -mean Q(s, pi(s)) (:168): per-row loss terms and the two output deltas
            if (row_thread) {
                const int r = tid;
                float *rt = lds + a.off_RT;
                const float y = rt[RT_R * kWR + r] + (1.0f - rt[RT_T * kWR + r]) * a.gamma * rt[RT_QT * kWR + r];
                const float e = rt[RT_Q * kWR + r] - y;
                const float inv_b = 1.0f / (float)a.batch;
                rt[RT_Y * kWR + r] = y;
                rt[RT_DQ * kWR + r] = valid ? 2.0f * e * inv_b : 0.0f;
                rt[RT_DQB * kWR + r] = valid ? -inv_b : 0.0f;
                float lc = valid ? e * e : 0.0f, la = valid ? -rt[RT_QPI * kWR + r] : 0.0f;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    lc += __shfl_xor(lc, m);
                    la += __shfl_xor(la, m);
                }
                if (r == 0) {
                    a.lpart[blockIdx.x * 2 + 0] = lc;
                    a.lpart[blockIdx.x * 2 + 1] = la;
                }
            }
            __syncthreads();
        }
        {
            const int l_first = __builtin_amdgcn_readfirstlane(T->ln_first[lv]), l_end = __builtin_amdgcn_readfirstlane(T->ln_first[lv + 1]);
            if (l_first < l_end) ln_level(T, l_first, l_end, lds, gout, tid);
        }
#ifdef SSC_WIDE_DIAG
        stamp[n_stamp++] = __builtin_amdgcn_s_memtime();
#endif
    }
#ifdef SSC_WIDE_DIAG
    }
#endif
    // ---- this workgroup's share of every gradient ----
    int job = wave;
    for (int wi = 0; wi < a.n_wg; ++wi) {
        const WGrad w = lds_uniform(&T->wg[wi]);
        const int tk = (w.in + 15) >> 4, tu = (w.out + 15) >> 4;
        const int tiles = tk * tu;
        for (; job < tiles; job += kWWaves) wgrad_tile(w, lds, gout, job / tu, job % tu, lane);
        job -= tiles;
    }
    for (int wi = 0; wi < a.n_wg; ++wi) {
        const WGrad w = lds_uniform(&T->wg[wi]);
        for (int u = tid; u < w.out; u += kWThreads) {
            const f4 *z = reinterpret_cast<const f4 *>(lds + w.dz_off + u * kWR);
            const f4 s4 = (z[0] + z[1]) + (z[2] + z[3]);
            gout[w.gb + u] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
    }
#ifdef SSC_WIDE_DIAG
    __syncthreads();
    stamp[n_stamp++] = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && tid == 0) {
        uint64_t *out = reinterpret_cast<uint64_t *>(a.lpart + 64);
        out[0] = (uint64_t)n_stamp;
        for (int i = 0; i < n_stamp; ++i) out[1 + i] = stamp[i];
    }
#endif
}

struct ApplyArgs {
    ssc_ddpg_desc d;
    const float *gpart, *lpart;
    float *losses;      // [2] of this iteration or nullptr
    int32_t n_blocks, nA, nC, it;
    // critic_l2_reg / clip_norm (ddpg_editted.py:175, 183-197): the summed gradient goes through `gsum` (prepare kernel)
    float *gsum;        // [nA + nC]
    float *regpart;     // [apply blocks] sum of squares of the block's regularised critic weights
    int32_t reg_lo[3], reg_hi[3];   // flat [actor | critic] ranges of the critic's dense kernels W1, W2, W3
    int32_t n_var;
    int32_t var_lo[21];             // variable boundaries in the flat order (clip_norm is per variable); var_lo[n_var] = nA + nC
};

// fixed-order sum of the per-workgroup partials; 32 loads in flight (the sum is a chain of L2 round trips: at 8 in
// flight a batch of 1024 -- 64 partials -- cost eight of them, 10.6 us per iteration in the actor-learner loop)
__device__ __forceinline__ float wide_partial_sum(const ApplyArgs &a, int p, int n) {
    float g = 0.0f;
    const float *src = a.gpart + p;
    int b = 0;
    for (; b + 32 <= a.n_blocks; b += 32) {
        float v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = src[(int64_t)(b + j) * n];
#pragma unroll
        for (int j = 0; j < 32; ++j) g += v[j];
    }
    for (; b + 8 <= a.n_blocks; b += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(b + j) * n];
#pragma unroll
        for (int j = 0; j < 8; ++j) g += v[j];
    }
    for (; b < a.n_blocks; ++b) g += src[(int64_t)b * n];
    return g;
}

// sum over the 256 threads of a block, the same order every time (xor butterfly inside a wave, then the four waves in order)
__device__ __forceinline__ float block256_sum(float x, float *red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    __syncthreads();                        // `red` may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// critic_l2_reg / clip_norm only: gsum = sum of the partials (+ critic_l2_reg * W on the critic's dense kernels: the
// gradient of scale * l2_loss(W), ddpg_editted.py:183-191), and the block's share of sum W^2 for the reported loss
__global__ __launch_bounds__(256) void ddpg_wide_prepare_kernel(ApplyArgs a) {
    __shared__ float red[4];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int n = a.nA + a.nC;
    float sq = 0.0f;
    if (p < n) {
        float g = wide_partial_sum(a, p, n);
        bool reg = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) reg = reg || (p >= a.reg_lo[k] && p < a.reg_hi[k]);
        if (reg && a.d.critic_l2_reg != 0.0f) {
            const float w = a.d.critic[p - a.nA];
            g += a.d.critic_l2_reg * w;
            sq = w * w;
        }
        a.gsum[p] = g;
    }
    sq = block256_sum(sq, red);
    if (threadIdx.x == 0) a.regpart[blockIdx.x] = sq;
}

template <bool PREPARED>
__global__ __launch_bounds__(256) void ddpg_wide_apply_kernel(ApplyArgs a) {
    __shared__ AdamCfg cfg[2];
    __shared__ float red[4];
    if (threadIdx.x < 2) {
        // MpiAdam's bias-corrected step size in f64 (1 - 0.999^t loses 5 digits in fp32); t = counter before this
        // call + iterations done + 1 (the counters themselves move once, after the last iteration)
        const int net = threadIdx.x;
        const int t = a.d.adam_t[net] + a.it + 1;
        const double b1 = ipow((double)a.d.beta1, t), b2 = ipow((double)a.d.beta2, t);
        const double lr = net == 0 ? (double)a.d.actor_lr : (double)a.d.critic_lr;
        cfg[net] = AdamCfg{(float)(lr * sqrt(1.0 - b2) / (1.0 - b1)), a.d.beta1, a.d.beta2, a.d.epsilon};
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int n = a.nA + a.nC;
    float g = 0.0f;
    if constexpr (PREPARED) {
        if (p < n) g = a.gsum[p];
        if (a.d.clip_norm > 0.0f) {
            // tf.clip_by_norm per variable (U.flatgrad(..., clip_norm), ddpg_editted.py:175, 197): every block sums the
            // squares of the variables its 256 parameters belong to (at most a few), in the same order in every block
            auto var_of = [&](int q) { int v = 0; while (v + 1 < a.n_var && q >= a.var_lo[v + 1]) ++v; return v; };
            const int last = (blockIdx.x * 256 + 255 < n ? blockIdx.x * 256 + 255 : n - 1);
            const int v_first = var_of(blockIdx.x * 256), v_last = var_of(last), mine = p < n ? var_of(p) : -1;
            float scale = 1.0f;
            for (int v = v_first; v <= v_last; ++v) {
                const int lo = a.var_lo[v], hi = a.var_lo[v + 1];
                float s = 0.0f;
                for (int q = lo + (int)threadIdx.x; q < hi; q += 256) { const float x = a.gsum[q]; s += x * x; }
                const float norm = sqrtf(block256_sum(s, red));
                if (mine == v) scale = a.d.clip_norm / fmaxf(norm, a.d.clip_norm);
            }
            g *= scale;
        }
    } else {
        if (p < n) g = wide_partial_sum(a, p, n);
    }
    if (p < n) {
        const int net = p < a.nA ? 0 : 1;
        const int q = net == 0 ? p : p - a.nA;
        float *theta = net == 0 ? a.d.actor : a.d.critic, *target = net == 0 ? a.d.target_actor : a.d.target_critic;
        float *mm = net == 0 ? a.d.adam_m_actor : a.d.adam_m_critic, *vv = net == 0 ? a.d.adam_v_actor : a.d.adam_v_critic;
        const AdamCfg c = cfg[net];
        // MpiAdam.update (baselines common/mpi_adam.py [third-party], ddpg_editted.py:326-327), update_target_net (:338-339)
        float m = mm[q], v = vv[q], th = theta[q], tg = target[q];
        m = c.beta1 * m + (1.0f - c.beta1) * g;
        v = c.beta2 * v + (1.0f - c.beta2) * (g * g);
        th += (-c.a) * m / (sqrtf(v) + c.eps);
        tg = (1.0f - a.d.tau) * tg + a.d.tau * th;
        mm[q] = m; vv[q] = v; theta[q] = th; target[q] = tg;
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 && a.losses != nullptr) {
        float s = 0.0f;
        for (int b = 0; b < a.n_blocks; ++b) s += a.lpart[b * 2 + threadIdx.x];
        s /= (float)a.d.batch_size;
        if (PREPARED && threadIdx.x == 0) {
            float w2 = 0.0f;
            for (unsigned b = 0; b < gridDim.x; ++b) w2 += a.regpart[b];
            s += 0.5f * a.d.critic_l2_reg * w2;
        }
        a.losses[threadIdx.x] = s;
    }
}

__global__ void ddpg_wide_finish_kernel(int32_t *adam_t, int32_t n_iters) {
    adam_t[0] += n_iters;
    adam_t[1] += n_iters;
}

struct WNet {
    int in, h1, h2, out, extra;   // extra: rows concatenated to the first hidden layer (critic: act_dim)
    int ln;                       // 1: [W1|b1|beta1|gamma1|W2|b2|beta2|gamma2|W3|b3] (ssc_ddpg_desc::layer_norm)
    int oW1() const { return 0; }
    int ob1() const { return in * h1; }
    int obe1() const { return ob1() + h1; }           // LayerNorm beta, then gamma (ln only)
    int og1() const { return obe1() + h1; }
    int oW2() const { return ob1() + h1 + (ln ? 2 * h1 : 0); }
    int ob2() const { return oW2() + (h1 + extra) * h2; }
    int obe2() const { return ob2() + h2; }
    int og2() const { return obe2() + h2; }
    int oW3() const { return ob2() + h2 + (ln ? 2 * h2 : 0); }
    int ob3() const { return oW3() + h2 * out; }
    int total() const { return ob3() + out; }
};

static int wide_blocks(const ssc_ddpg_desc *d) { return (d->batch_size + kWR - 1) / kWR; }
static int wide_params(const ssc_ddpg_desc *d) {
    const int ln = d->layer_norm ? 1 : 0;
    const WNet A{d->obs_dim, d->actor_h1, d->actor_h2, d->act_dim, 0, ln}, C{d->obs_dim, d->critic_h1, d->critic_h2, 1, d->act_dim, ln};
    return A.total() + C.total();
}

static bool wide_prepared(const ssc_ddpg_desc *d) { return d->critic_l2_reg != 0.0f || d->clip_norm > 0.0f; }
static size_t wide_gpart_bytes(const ssc_ddpg_desc *d, int nb) { return ((size_t)nb * (size_t)wide_params(d) * sizeof(float) + 255) & ~(size_t)255; }
static size_t wide_lpart_bytes(int nb) { return (((size_t)nb * 2 * sizeof(float) + 255) & ~(size_t)255) + 1024; }

// [gpart | lpart (+ 1 KB the diagnostic build stamps into)] and, with critic_l2_reg / clip_norm, [gsum | regpart]; sized
// for 16-row workgroups (the 64-row tiles of ddpg_train_fixed_tiled lay out fewer partials in the same workspace)
size_t ddpg_wide_workspace_bytes(const ssc_ddpg_desc *d) {
    size_t bytes = wide_gpart_bytes(d, wide_blocks(d)) + wide_lpart_bytes(wide_blocks(d));
    if (wide_prepared(d)) {
        const size_t n = (size_t)wide_params(d);
        bytes += ((n * sizeof(float) + 255) & ~(size_t)255) + (((n + 255) / 256 * sizeof(float) + 255) & ~(size_t)255);
    }
    return bytes;
}

WidePartials ddpg_wide_partials(const ssc_ddpg_desc *d, void *d_workspace, int n_blocks) {
    return WidePartials{static_cast<float *>(d_workspace),
                        reinterpret_cast<float *>(static_cast<char *>(d_workspace) + wide_gpart_bytes(d, n_blocks))};
}

void ddpg_wide_apply(const ssc_ddpg_desc *d, void *d_workspace, int n_blocks, int it, float *d_losses_it, hipStream_t stream) {
    const int ln = d->layer_norm ? 1 : 0;
    const WNet A{d->obs_dim, d->actor_h1, d->actor_h2, d->act_dim, 0, ln}, C{d->obs_dim, d->critic_h1, d->critic_h2, 1, d->act_dim, ln};
    const int nA = A.total(), nC = C.total();
    const WidePartials wp = ddpg_wide_partials(d, d_workspace, n_blocks);
    ApplyArgs ap{};
    ap.d = *d; ap.gpart = wp.gpart; ap.lpart = wp.lpart; ap.n_blocks = n_blocks; ap.nA = nA; ap.nC = nC;
    ap.it = it; ap.losses = d_losses_it;
    const unsigned apply_blocks = (unsigned)((nA + nC + 255) / 256);
    if (!wide_prepared(d)) {
        hipLaunchKernelGGL(ddpg_wide_apply_kernel<false>, dim3(apply_blocks), dim3(256), 0, stream, ap);
        return;
    }
    char *tail = static_cast<char *>(d_workspace) + wide_gpart_bytes(d, n_blocks) + wide_lpart_bytes(n_blocks);
    ap.gsum = reinterpret_cast<float *>(tail);
    ap.regpart = reinterpret_cast<float *>(tail + (((size_t)(nA + nC) * sizeof(float) + 255) & ~(size_t)255));
    // the name filter of ddpg_editted.py:184 ('kernel' in name, 'output' not in name) keeps all three dense kernels of
    // models_editted.py's critic (none of its layers is named 'output'); biases and LayerNorm parameters stay out
    const int lo[3] = {nA + C.oW1(), nA + C.oW2(), nA + C.oW3()}, hi[3] = {nA + C.ob1(), nA + C.ob2(), nA + C.ob3()};
    for (int k = 0; k < 3; ++k) { ap.reg_lo[k] = lo[k]; ap.reg_hi[k] = hi[k]; }
    int nv = 0;
    for (int n = 0; n < 2; ++n) {
        const WNet &N = n == 0 ? A : C;
        const int base = n == 0 ? 0 : nA;
        ap.var_lo[nv++] = base + N.oW1(); ap.var_lo[nv++] = base + N.ob1();
        if (ln) { ap.var_lo[nv++] = base + N.obe1(); ap.var_lo[nv++] = base + N.og1(); }
        ap.var_lo[nv++] = base + N.oW2(); ap.var_lo[nv++] = base + N.ob2();
        if (ln) { ap.var_lo[nv++] = base + N.obe2(); ap.var_lo[nv++] = base + N.og2(); }
        ap.var_lo[nv++] = base + N.oW3(); ap.var_lo[nv++] = base + N.ob3();
    }
    ap.n_var = nv;
    ap.var_lo[nv] = nA + nC;
    hipLaunchKernelGGL(ddpg_wide_prepare_kernel, dim3(apply_blocks), dim3(256), 0, stream, ap);
    hipLaunchKernelGGL(ddpg_wide_apply_kernel<true>, dim3(apply_blocks), dim3(256), 0, stream, ap);
}

void ddpg_wide_finish(const ssc_ddpg_desc *d, int32_t n_iters, hipStream_t stream) {
    hipLaunchKernelGGL(ddpg_wide_finish_kernel, dim3(1), dim3(1), 0, stream, d->adam_t, n_iters);
}

int ddpg_train_wide(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                    float *d_losses, void *d_workspace, size_t workspace_bytes, hipStream_t stream) {
    if (d->batch_size < 1 || d->batch_size > 4096)
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: batch_size %d not in 1..4096", d->batch_size);
    const size_t need = ddpg_wide_workspace_bytes(d);
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= need,
                "ssc_ddpg_train_ws: workspace %zu < %zu bytes (ssc_ddpg_train_workspace_bytes)", workspace_bytes, need);
    const int ln = d->layer_norm ? 1 : 0;
    const WNet A{d->obs_dim, d->actor_h1, d->actor_h2, d->act_dim, 0, ln}, C{d->obs_dim, d->critic_h1, d->critic_h2, 1, d->act_dim, ln};
    const int od = d->obs_dim, ad = d->act_dim;
    const int act2 = d->last_layer_tanh ? EPI_TANH : EPI_RELU, mask2 = d->last_layer_tanh ? EPI_MASK_TANH : EPI_MASK_RELU;
    // with LayerNorm the contractions of the hidden layers leave the pre-normalisation sums and the LayerNorm op applies
    // the activation (ln_level)
    const int epi1 = ln ? EPI_NONE : EPI_RELU, epi2 = ln ? EPI_NONE : act2;
    WideArgs g{};
    g.rp = *rp; g.batch = d->batch_size; g.obs_dim = od; g.act_dim = ad; g.gamma = d->gamma; g.obs_clip = d->obs_clip;
    // ---- LDS carve: rows of 16 floats ([unit][row]) ------------------------------------------------------------------
    int p = 0;
    auto take = [&](int rows) { const int q = p; p += rows * kWR; return q; };
    const int S = take(od), S2 = take(od), RT = take(RT_ROWS);
    const int TA1 = take(A.h1), TA2 = take(A.h2);          // target actor; later the actor's deltas dU1, dU2
    const int TC1 = take(C.h1 + ad);                       // target critic layer 1 ++ target action; later dZ1
    const int TZ2 = take(C.h2);                            // target critic layer 2; later dZ2
    const int C1 = take(C.h1 + ad);                        // critic layer 1 ++ the batch's action rows
    const int Z2H = take(C.h2);                            // W2[:h1]^T relu(layer 1) + b2, shared by Q(s, a) and Q(s, pi(s)); later dZB2
    const int C2 = take(C.h2), CB2 = take(C.h2);
    const int U1 = take(A.h1), U2 = take(A.h2), PI = take(ad), DPI = take(ad);
    // LayerNorm: x-hat of every normalised layer a gradient flows through, one 1/sigma row each, and the reduction scratch
    int XC1 = -1, XC2 = -1, XCB2 = -1, XU1 = -1, XU2 = -1, RS = -1;
    if (ln) {
        XC1 = take(C.h1); XC2 = take(C.h2); XCB2 = take(C.h2); XU1 = take(A.h1); XU2 = take(A.h2);
        RS = take(5);
        g.tab.off_scr = p;
        p += kLnPerLevel * 2 * kLnParts * kWR;
    }
    const int DU1 = TA1, DU2 = TA2, DZ1 = TC1, DZ2 = TZ2, DZB2 = Z2H;
    const int TPI = TC1 + C.h1 * kWR, ACT = C1 + C.h1 * kWR;
    g.off_S = S; g.off_S2 = S2; g.off_ACT = ACT; g.off_TACT = TPI; g.off_RT = RT;
    // the small-parameter image: per net the flat vector without its h1 x h2 block, i.e. [W1 | b1] and
    // [W2 action rows | b2 | W3 | b3]; img(net, flat offset) = LDS offset of that element
    const float *theta[4] = {d->target_actor, d->target_critic, d->critic, d->actor};     // ta, tc, c, a
    const WNet *nets[4] = {&A, &C, &C, &A};
    int img_lo[4], img_hi[4];
    g.tab.n_seg = 0; g.tab.n_big = 0;
    for (int n = 0; n < 4; ++n) {
        const WNet &N = *nets[n];
        const int head = N.h1 * N.h2;
        img_lo[n] = p;
        g.tab.seg[g.tab.n_seg].src = theta[n]; g.tab.seg[g.tab.n_seg].dst = p; g.tab.seg[g.tab.n_seg].n = N.oW2();
        ++g.tab.n_seg;
        p += N.oW2();
        img_hi[n] = p - (N.oW2() + head);        // flat offsets >= oW2 + head map to img_hi + offset
        g.tab.seg[g.tab.n_seg].src = theta[n] + N.oW2() + head; g.tab.seg[g.tab.n_seg].dst = p; g.tab.seg[g.tab.n_seg].n = N.total() - N.oW2() - head;
        ++g.tab.n_seg;
        p += N.total() - N.oW2() - head;
        g.tab.big[g.tab.n_big].src = theta[n] + N.oW2(); g.tab.big[g.tab.n_big].n = head;
        ++g.tab.n_big;
    }
    auto img = [&](int n, int flat) { return flat < nets[n]->oW2() ? img_lo[n] + flat : img_hi[n] + flat; };
    g.off_tab = (p + 3) & ~3;
    p = g.off_tab + (int)(sizeof(WideTables) / 4);
    enum { TA = 0, TC = 1, CR = 2, AC = 3 };
    const size_t lds = (size_t)p * sizeof(float);
    if (lds > 160 * 1024)
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: these layer sizes need %zu B of LDS per workgroup (160 KB available)", lds);
    int ng = 0, nl = 0, nln = 0;
    auto level = [&]() { g.tab.ln_first[nl] = nln; g.tab.level_first[nl++] = ng; };
    // LayerNorm ops behind the current level's contractions; net n's gamma / beta of hidden layer `layer`
    auto ln_fwd = [&](int n, int layer, int x_off, int act, int xhat_off, int rstd_off) {
        if (!ln) return;
        const WNet &N = *nets[n];
        WLn &L = g.tab.ln[nln++];
        L = WLn{0, x_off, xhat_off, rstd_off, img(n, layer == 1 ? N.og1() : N.og2()), img(n, layer == 1 ? N.obe1() : N.obe2()),
                layer == 1 ? N.h1 : N.h2, act, -1, -1};
    };
    auto ln_bwd = [&](int n, int layer, int x_off, int xhat_off, int rstd_off, int gbase) {   // gbase < 0: no parameter gradients
        if (!ln) return;
        const WNet &N = *nets[n];
        WLn &L = g.tab.ln[nln++];
        L = WLn{1, x_off, xhat_off, rstd_off, img(n, layer == 1 ? N.og1() : N.og2()), -1, layer == 1 ? N.h1 : N.h2, EPI_NONE,
                gbase < 0 ? -1 : gbase + (layer == 1 ? N.og1() : N.og2()), gbase < 0 ? -1 : gbase + (layer == 1 ? N.obe1() : N.obe2())};
    };
    // net n, flat offset wflat of A(0, 0); big = true: the h1 x h2 block, streamed from L2
    auto gemm = [&](int n, int wflat, bool big, int sm, int sk, int M, int K, int bias_flat, int b_off, int out_off, int epi,
                    int add_off = -1, int aux_off = -1) {
        WGemm &x = g.tab.gemm[ng++];
        x.W = big ? theta[n] + wflat : nullptr;
        x.w_off = big ? 0 : img(n, wflat);
        x.bias_off = bias_flat >= 0 ? img(n, bias_flat) : -1;
        x.sm = sm; x.sk = sk; x.M = M; x.K = K; x.b_off = b_off; x.out_off = out_off; x.epi = epi;
        x.add_off = add_off; x.aux_off = aux_off;
    };
    const int RQ = RT + RT_Q * kWR, RQT = RT + RT_QT * kWR, RQPI = RT + RT_QPI * kWR, RDQ = RT + RT_DQ * kWR, RDQB = RT + RT_DQB * kWR;
    const int nA_ = A.total();
    // L0: the first layer of all four networks
    level();
    gemm(TA, A.oW1(), false, 1, A.h1, A.h1, od, A.ob1(), S2, TA1, epi1);
    gemm(TC, C.oW1(), false, 1, C.h1, C.h1, od, C.ob1(), S2, TC1, epi1);
    gemm(CR, C.oW1(), false, 1, C.h1, C.h1, od, C.ob1(), S, C1, epi1);
    gemm(AC, A.oW1(), false, 1, A.h1, A.h1, od, A.ob1(), S, U1, epi1);
    ln_fwd(TA, 1, TA1, EPI_RELU, -1, -1);
    ln_fwd(TC, 1, TC1, EPI_RELU, -1, -1);
    ln_fwd(CR, 1, C1, EPI_RELU, XC1, RS + 0 * kWR);
    ln_fwd(AC, 1, U1, EPI_RELU, XU1, RS + 3 * kWR);
    // L1: the h1 x h2 contractions; the critics' second layer without its action rows (they need pi' / pi)
    level();
    gemm(TA, A.oW2(), true, 1, A.h2, A.h2, A.h1, A.ob2(), TA1, TA2, epi2);
    gemm(AC, A.oW2(), true, 1, A.h2, A.h2, A.h1, A.ob2(), U1, U2, epi2);
    gemm(CR, C.oW2(), true, 1, C.h2, C.h2, C.h1, C.ob2(), C1, Z2H, EPI_NONE);
    gemm(TC, C.oW2(), true, 1, C.h2, C.h2, C.h1, C.ob2(), TC1, TZ2, EPI_NONE);
    ln_fwd(TA, 2, TA2, act2, -1, -1);
    ln_fwd(AC, 2, U2, act2, XU2, RS + 4 * kWR);
    // L2: pi'(s2), pi(s), Q(s, a) layer 2 = act(head + W2[h1:]^T a)
    level();
    gemm(TA, A.oW3(), false, 1, ad, ad, A.h2, A.ob3(), TA2, TPI, EPI_TANH);
    gemm(AC, A.oW3(), false, 1, ad, ad, A.h2, A.ob3(), U2, PI, EPI_TANH);
    gemm(CR, C.oW2() + C.h1 * C.h2, false, 1, C.h2, C.h2, ad, -1, ACT, C2, epi2, Z2H);
    ln_fwd(CR, 2, C2, act2, XC2, RS + 1 * kWR);
    // L3: Q'(s2, pi') layer 2, Q(s, pi(s)) layer 2, Q(s, a)
    level();
    gemm(TC, C.oW2() + C.h1 * C.h2, false, 1, C.h2, C.h2, ad, -1, TPI, TZ2, epi2, TZ2);
    gemm(CR, C.oW2() + C.h1 * C.h2, false, 1, C.h2, C.h2, ad, -1, PI, CB2, epi2, Z2H);
    gemm(CR, C.oW3(), false, 1, 1, 1, C.h2, C.ob3(), C2, RQ, EPI_NONE);
    ln_fwd(TC, 2, TZ2, act2, -1, -1);
    ln_fwd(CR, 2, CB2, act2, XCB2, RS + 2 * kWR);
    // L4: Q'(s2, pi'(s2)) and Q(s, pi(s)); then target_Q, the losses and the output deltas
    level();
    gemm(TC, C.oW3(), false, 1, 1, 1, C.h2, C.ob3(), TZ2, RQT, EPI_NONE);
    gemm(CR, C.oW3(), false, 1, 1, 1, C.h2, C.ob3(), CB2, RQPI, EPI_NONE);
    g.td_level = nl - 1;
    // L5: dz2 = (W3 dq) * act'(z2) for the critic loss, the same through Q(s, pi(s)) for the actor loss
    // (with LayerNorm the contraction leaves dL/d(LayerNorm output) and the op behind it turns it into dL/dz; the actor-loss
    // path goes through the critic's normalisation too but leaves its gamma / beta alone)
    level();
    gemm(CR, C.oW3(), false, 1, 1, C.h2, 1, -1, RDQ, DZ2, mask2, -1, C2);
    gemm(CR, C.oW3(), false, 1, 1, C.h2, 1, -1, RDQB, DZB2, mask2, -1, CB2);
    ln_bwd(CR, 2, DZ2, XC2, RS + 1 * kWR, nA_);
    ln_bwd(CR, 2, DZB2, XCB2, RS + 2 * kWR, -1);
    // L6: critic layer-1 deltas; d(-mean Q)/d(action) through the actor's output tanh
    level();
    gemm(CR, C.oW2(), true, C.h2, 1, C.h1, C.h2, -1, DZ2, DZ1, EPI_MASK_RELU, -1, C1);
    gemm(CR, C.oW2() + C.h1 * C.h2, false, C.h2, 1, ad, C.h2, -1, DZB2, DPI, EPI_MASK_TANH, -1, PI);
    ln_bwd(CR, 1, DZ1, XC1, RS + 0 * kWR, nA_);
    // L7, L8: back through the actor
    level();
    gemm(AC, A.oW3(), false, ad, 1, A.h2, ad, -1, DPI, DU2, mask2, -1, U2);
    ln_bwd(AC, 2, DU2, XU2, RS + 4 * kWR, 0);
    level();
    gemm(AC, A.oW2(), true, A.h2, 1, A.h1, A.h2, -1, DU2, DU1, EPI_MASK_RELU, -1, U1);
    ln_bwd(AC, 1, DU1, XU1, RS + 3 * kWR, 0);
    g.tab.ln_first[nl] = nln;
    g.tab.level_first[nl] = ng;
    g.n_gemm = ng; g.n_level = nl;
    // gradients, flat [actor | critic], TF trainable_vars order inside a net
    const int nA = A.total(), nC = C.total();
    int nw = 0;
    auto wgrad = [&](int x_off, int dz_off, int in, int out, int gW, int gb) { g.tab.wg[nw++] = WGrad{x_off, dz_off, in, out, gW, gb}; };
    wgrad(C1, DZ2, C.h1 + ad, C.h2, nA + C.oW2(), nA + C.ob2());       // the large ones first: they are dealt out first
    wgrad(U1, DU2, A.h1, A.h2, A.oW2(), A.ob2());
    wgrad(S, DZ1, od, C.h1, nA + C.oW1(), nA + C.ob1());
    wgrad(S, DU1, od, A.h1, A.oW1(), A.ob1());
    wgrad(C2, RDQ, C.h2, 1, nA + C.oW3(), nA + C.ob3());
    wgrad(U2, DPI, A.h2, ad, A.oW3(), A.ob3());
    g.n_wg = nw;
    g.n_params = nA + nC;
    const int nb = wide_blocks(d);
    g.gpart = static_cast<float *>(d_workspace);
    g.lpart = ddpg_wide_partials(d, d_workspace, nb).lpart;
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(ddpg_wide_grad_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(ddpg_wide_grad_kernel)");
        if (rc) return rc;
    }
    for (int it = 0; it < n_iters; ++it) {
        g.batch_idx = d_batch_idx + (int64_t)it * d->batch_size;
        hipLaunchKernelGGL(ddpg_wide_grad_kernel, dim3(nb), dim3(kWThreads), lds, stream, g);
        ddpg_wide_apply(d, d_workspace, nb, it, d_losses ? d_losses + 2 * it : nullptr, stream);
    }
    ddpg_wide_finish(d, n_iters, stream);
    return check_launch("ssc_ddpg_train (wide)");
}

}  // namespace ssc
