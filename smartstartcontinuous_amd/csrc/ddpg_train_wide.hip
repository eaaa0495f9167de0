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
constexpr int kMaxGemm = 24, kMaxLevel = 12, kMaxWg = 12;

enum : int { EPI_NONE = 0, EPI_RELU, EPI_TANH, EPI_MASK_RELU, EPI_MASK_TANH };

// out[m][row] = EPI( sum_kk A(m, kk) * B[kk][row] + bias[m] + add[m][row] ),  A(m, kk) = W[m * sm + kk * sk]
struct WGemm {
    const float *W;      // global (parameters; read-only during the launch)
    const float *bias;   // global [M] or nullptr
    int32_t sm, sk, M, K;
    int32_t b_off, out_off, add_off, aux_off;   // LDS float offsets; add_off / aux_off < 0: none
    int32_t epi;
};

// gradient of one layer: dW[k][u] = sum_rows X[k][row] * dZ[u][row], db[u] = sum_rows dZ[u][row]
struct WGrad {
    int32_t x_off, dz_off, in, out;   // LDS float offsets of the X rows [in][16] and the delta rows [out][16]
    int32_t gW, gb;                   // offsets into the flat gradient vector (actor first, then critic)
};

struct WideArgs {
    ssc_replay_view rp;
    const int32_t *batch_idx;    // [batch] of this iteration
    int32_t batch, obs_dim, act_dim;
    float gamma, obs_clip;
    int32_t off_S, off_S2, off_ACT, off_TACT, off_RT;   // LDS float offsets; RT: rows r, t, y, q, q', qpi, dq, dqb
    int32_t n_gemm, n_level, td_level, n_wg;
    int32_t level_first[kMaxLevel + 1];
    WGemm gemm[kMaxGemm];
    WGrad wg[kMaxWg];
    float *gpart;   // [n_blocks][n_params]
    float *lpart;   // [n_blocks][2]
    int32_t n_params;
};

enum : int { RT_R = 0, RT_T, RT_Y, RT_Q, RT_QT, RT_QPI, RT_DQ, RT_DQB, RT_ROWS };

// One job of a contraction: up to two 16-unit tiles (they share the B operand) for the workgroup's 16 rows.
__device__ __forceinline__ void gemm_job(const WGemm &g, float *lds, int tile0, int lane) {
    constexpr int KU = 8;
    const int lm = lane & 15, lk = lane >> 4;
    const int m0 = tile0 * 16 + lm, m1 = m0 + 16;
    const bool v0 = m0 < g.M, v1 = m1 < g.M;
    const float *w0 = g.W + (int64_t)m0 * g.sm + (int64_t)lk * g.sk;
    const float *w1 = w0 + (int64_t)16 * g.sm;
    const float *b = lds + g.b_off + lk * kWR + lm;
    f32x4m acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int steps = (g.K + 3) >> 2;
    for (int s0 = 0; s0 < steps; s0 += KU) {
        float a0[KU], a1[KU], bv[KU];
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            const int kk = 4 * (s0 + j) + lk;
            const bool ok = kk < g.K;
            a0[j] = (ok && v0) ? w0[(int64_t)4 * (s0 + j) * g.sk] : 0.0f;
            a1[j] = (ok && v1) ? w1[(int64_t)4 * (s0 + j) * g.sk] : 0.0f;
            bv[j] = ok ? b[(s0 + j) * 4 * kWR] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < KU; ++j) {
            acc0 = mfma4(a0[j], bv[j], acc0);
            acc1 = mfma4(a1[j], bv[j], acc1);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4m &acc = t == 0 ? acc0 : acc1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = (tile0 + t) * 16 + 4 * lk + i;
            if (m >= g.M) continue;
            float v = acc[i];
            if (g.bias != nullptr) v += g.bias[m];
            if (g.add_off >= 0) v += lds[g.add_off + m * kWR + lm];
            if (g.epi == EPI_RELU) v = fmaxf(v, 0.0f);
            else if (g.epi == EPI_TANH) v = tanhf(v);
            else if (g.epi == EPI_MASK_RELU) v = lds[g.aux_off + m * kWR + lm] > 0.0f ? v : 0.0f;
            else if (g.epi == EPI_MASK_TANH) { const float a = lds[g.aux_off + m * kWR + lm]; v *= 1.0f - a * a; }
            lds[g.out_off + m * kWR + lm] = v;
        }
    }
}

// One 16 x 16 tile of a layer's weight gradient (tile i over the inputs, tile j over the units): the 16 rows are the
// K of four MFMAs; lane group lk contracts rows 4 lk .. 4 lk + 3 (any pairing of k values is a valid contraction),
// so both operands come in with one 16-byte LDS read per lane.
__device__ __forceinline__ void wgrad_tile(const WGrad &w, const float *lds, float *gout, int ti, int tj, int lane) {
    const int lm = lane & 15, lk = lane >> 4;
    const int k = ti * 16 + lm, u = tj * 16 + lm;
    f4 xa = {0.0f, 0.0f, 0.0f, 0.0f}, dz = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k < w.in) xa = *reinterpret_cast<const f4 *>(lds + w.x_off + k * kWR + 4 * lk);
    if (u < w.out) dz = *reinterpret_cast<const f4 *>(lds + w.dz_off + u * kWR + 4 * lk);
    f32x4m acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma4(xa[s], dz[s], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kk = ti * 16 + 4 * lk + i;
        if (kk < w.in && u < w.out) gout[w.gW + kk * w.out + u] = acc[i];
    }
}

__global__ __launch_bounds__(kWThreads) void ddpg_wide_grad_kernel(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * kWR;
    // ---- ReplayBuffer.sample_batch rows of this workgroup (replay_buffer.py:79-91); rows past the batch shadow its
    // last record and carry zero weight in every loss ----
    if (tid < kWR) {
        const int r = tid;
        const bool valid = row0 + r < a.batch;
        const int64_t rec = a.batch_idx[valid ? row0 + r : a.batch - 1];
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
    float *gout = a.gpart + (int64_t)blockIdx.x * a.n_params;
    for (int lv = 0; lv < a.n_level; ++lv) {
        // the level's independent contractions as (contraction, tile-pair) jobs, dealt round-robin to the waves
        int job = wave;
        for (int gi = a.level_first[lv]; gi < a.level_first[lv + 1]; ++gi) {
            const WGemm &g = a.gemm[gi];
            const int pairs = (((g.M + 15) >> 4) + 1) >> 1;
            while (job < pairs) {
                gemm_job(g, lds, 2 * job, lane);
                job += kWWaves;
            }
            job -= pairs;
        }
        __syncthreads();
        if (lv == a.td_level) {
            // target_Q = r + (1 - terminal) * gamma * Q'(s2, pi'(s2))  (:132-133); critic loss mean((Q - y)^2) (:181),
            // actor loss -mean Q(s, pi(s)) (:168): per-row loss terms and the two output deltas
            if (tid < kWR) {
                const int r = tid;
                float *rt = lds + a.off_RT;
                const bool valid = row0 + r < a.batch;
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
    }
    // ---- this workgroup's share of every gradient ----
    int job = wave;
    for (int wi = 0; wi < a.n_wg; ++wi) {
        const WGrad &w = a.wg[wi];
        const int tk = (w.in + 15) >> 4, tu = (w.out + 15) >> 4;
        const int tiles = tk * tu;
        while (job < tiles) {
            wgrad_tile(w, lds, gout, job / tu, job % tu, lane);
            job += kWWaves;
        }
        job -= tiles;
    }
    for (int wi = 0; wi < a.n_wg; ++wi) {
        const WGrad &w = a.wg[wi];
        for (int u = tid; u < w.out; u += kWThreads) {
            const f4 *z = reinterpret_cast<const f4 *>(lds + w.dz_off + u * kWR);
            const f4 s4 = (z[0] + z[1]) + (z[2] + z[3]);
            gout[w.gb + u] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
    }
}

struct ApplyArgs {
    ssc_ddpg_desc d;
    const float *gpart, *lpart;
    float *losses;      // [2] of this iteration or nullptr
    int32_t n_blocks, nA, nC, it;
};

__global__ __launch_bounds__(256) void ddpg_wide_apply_kernel(ApplyArgs a) {
    __shared__ AdamCfg cfg[2];
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
    if (p < n) {
        // fixed-order sum of the per-workgroup partials, 8 loads in flight
        float g = 0.0f;
        const float *src = a.gpart + p;
        int b = 0;
        for (; b + 8 <= a.n_blocks; b += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(b + j) * n];
#pragma unroll
            for (int j = 0; j < 8; ++j) g += v[j];
        }
        for (; b < a.n_blocks; ++b) g += src[(int64_t)b * n];
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
        a.losses[threadIdx.x] = s / (float)a.d.batch_size;
    }
}

__global__ void ddpg_wide_finish_kernel(int32_t *adam_t, int32_t n_iters) {
    adam_t[0] += n_iters;
    adam_t[1] += n_iters;
}

struct WNet {
    int in, h1, h2, out, extra;   // extra: rows concatenated to the first hidden layer (critic: act_dim)
    int oW1() const { return 0; }
    int ob1() const { return in * h1; }
    int oW2() const { return ob1() + h1; }
    int ob2() const { return oW2() + (h1 + extra) * h2; }
    int oW3() const { return ob2() + h2; }
    int ob3() const { return oW3() + h2 * out; }
    int total() const { return ob3() + out; }
};

static int wide_blocks(const ssc_ddpg_desc *d) { return (d->batch_size + kWR - 1) / kWR; }
static int wide_params(const ssc_ddpg_desc *d) {
    const WNet A{d->obs_dim, d->actor_h1, d->actor_h2, d->act_dim, 0}, C{d->obs_dim, d->critic_h1, d->critic_h2, 1, d->act_dim};
    return A.total() + C.total();
}

size_t ddpg_wide_workspace_bytes(const ssc_ddpg_desc *d) {
    const size_t nb = (size_t)wide_blocks(d);
    return ((nb * (size_t)wide_params(d) * sizeof(float) + 255) & ~(size_t)255) + ((nb * 2 * sizeof(float) + 255) & ~(size_t)255);
}

int ddpg_train_wide(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                    float *d_losses, void *d_workspace, size_t workspace_bytes, hipStream_t stream) {
    if (d->batch_size < 1 || d->batch_size > 4096)
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: batch_size %d not in 1..4096", d->batch_size);
    const size_t need = ddpg_wide_workspace_bytes(d);
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= need,
                "ssc_ddpg_train_ws: workspace %zu < %zu bytes (ssc_ddpg_train_workspace_bytes)", workspace_bytes, need);
    const WNet A{d->obs_dim, d->actor_h1, d->actor_h2, d->act_dim, 0}, C{d->obs_dim, d->critic_h1, d->critic_h2, 1, d->act_dim};
    const int od = d->obs_dim, ad = d->act_dim;
    const int act2 = d->last_layer_tanh ? EPI_TANH : EPI_RELU, mask2 = d->last_layer_tanh ? EPI_MASK_TANH : EPI_MASK_RELU;
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
    const int DU1 = TA1, DU2 = TA2, DZ1 = TC1, DZ2 = TZ2, DZB2 = Z2H;
    const int TPI = TC1 + C.h1 * kWR, ACT = C1 + C.h1 * kWR;
    g.off_S = S; g.off_S2 = S2; g.off_ACT = ACT; g.off_TACT = TPI; g.off_RT = RT;
    const size_t lds = (size_t)p * sizeof(float);
    if (lds > 160 * 1024)
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: these layer sizes need %zu B of LDS per workgroup (160 KB available)", lds);
    const float *ta = d->target_actor, *tc = d->target_critic, *th_a = d->actor, *th_c = d->critic;
    int ng = 0, nl = 0;
    auto level = [&]() { g.level_first[nl++] = ng; };
    auto gemm = [&](const float *W, int sm, int sk, int M, int K, const float *bias, int b_off, int out_off, int epi,
                    int add_off = -1, int aux_off = -1) {
        WGemm &x = g.gemm[ng++];
        x.W = W; x.sm = sm; x.sk = sk; x.M = M; x.K = K; x.bias = bias; x.b_off = b_off; x.out_off = out_off; x.epi = epi;
        x.add_off = add_off; x.aux_off = aux_off;
    };
    const int RQ = RT + RT_Q * kWR, RQT = RT + RT_QT * kWR, RQPI = RT + RT_QPI * kWR, RDQ = RT + RT_DQ * kWR, RDQB = RT + RT_DQB * kWR;
    // L0: the first layer of all four networks
    level();
    gemm(ta + A.oW1(), 1, A.h1, A.h1, od, ta + A.ob1(), S2, TA1, EPI_RELU);
    gemm(tc + C.oW1(), 1, C.h1, C.h1, od, tc + C.ob1(), S2, TC1, EPI_RELU);
    gemm(th_c + C.oW1(), 1, C.h1, C.h1, od, th_c + C.ob1(), S, C1, EPI_RELU);
    gemm(th_a + A.oW1(), 1, A.h1, A.h1, od, th_a + A.ob1(), S, U1, EPI_RELU);
    // L1: the h1 x h2 contractions; the critics' second layer without its action rows (they need pi' / pi)
    level();
    gemm(ta + A.oW2(), 1, A.h2, A.h2, A.h1, ta + A.ob2(), TA1, TA2, act2);
    gemm(th_a + A.oW2(), 1, A.h2, A.h2, A.h1, th_a + A.ob2(), U1, U2, act2);
    gemm(th_c + C.oW2(), 1, C.h2, C.h2, C.h1, th_c + C.ob2(), C1, Z2H, EPI_NONE);
    gemm(tc + C.oW2(), 1, C.h2, C.h2, C.h1, tc + C.ob2(), TC1, TZ2, EPI_NONE);
    // L2: pi'(s2), pi(s), Q(s, a) layer 2 = act(head + W2[h1:]^T a)
    level();
    gemm(ta + A.oW3(), 1, ad, ad, A.h2, ta + A.ob3(), TA2, TPI, EPI_TANH);
    gemm(th_a + A.oW3(), 1, ad, ad, A.h2, th_a + A.ob3(), U2, PI, EPI_TANH);
    gemm(th_c + C.oW2() + C.h1 * C.h2, 1, C.h2, C.h2, ad, nullptr, ACT, C2, act2, Z2H);
    // L3: Q'(s2, pi') layer 2, Q(s, pi(s)) layer 2, Q(s, a)
    level();
    gemm(tc + C.oW2() + C.h1 * C.h2, 1, C.h2, C.h2, ad, nullptr, TPI, TZ2, act2, TZ2);
    gemm(th_c + C.oW2() + C.h1 * C.h2, 1, C.h2, C.h2, ad, nullptr, PI, CB2, act2, Z2H);
    gemm(th_c + C.oW3(), 1, 1, 1, C.h2, th_c + C.ob3(), C2, RQ, EPI_NONE);
    // L4: Q'(s2, pi'(s2)) and Q(s, pi(s)); then target_Q, the losses and the output deltas
    level();
    gemm(tc + C.oW3(), 1, 1, 1, C.h2, tc + C.ob3(), TZ2, RQT, EPI_NONE);
    gemm(th_c + C.oW3(), 1, 1, 1, C.h2, th_c + C.ob3(), CB2, RQPI, EPI_NONE);
    g.td_level = nl - 1;
    // L5: dz2 = (W3 dq) * act'(z2) for the critic loss, the same through Q(s, pi(s)) for the actor loss
    level();
    gemm(th_c + C.oW3(), 1, 1, C.h2, 1, nullptr, RDQ, DZ2, mask2, -1, C2);
    gemm(th_c + C.oW3(), 1, 1, C.h2, 1, nullptr, RDQB, DZB2, mask2, -1, CB2);
    // L6: critic layer-1 deltas; d(-mean Q)/d(action) through the actor's output tanh
    level();
    gemm(th_c + C.oW2(), C.h2, 1, C.h1, C.h2, nullptr, DZ2, DZ1, EPI_MASK_RELU, -1, C1);
    gemm(th_c + C.oW2() + C.h1 * C.h2, C.h2, 1, ad, C.h2, nullptr, DZB2, DPI, EPI_MASK_TANH, -1, PI);
    // L7, L8: back through the actor
    level();
    gemm(th_a + A.oW3(), ad, 1, A.h2, ad, nullptr, DPI, DU2, mask2, -1, U2);
    level();
    gemm(th_a + A.oW2(), A.h2, 1, A.h1, A.h2, nullptr, DU2, DU1, EPI_MASK_RELU, -1, U1);
    g.level_first[nl] = ng;
    g.n_gemm = ng; g.n_level = nl;
    // gradients, flat [actor | critic], TF trainable_vars order inside a net
    const int nA = A.total(), nC = C.total();
    int nw = 0;
    auto wgrad = [&](int x_off, int dz_off, int in, int out, int gW, int gb) { g.wg[nw++] = WGrad{x_off, dz_off, in, out, gW, gb}; };
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
    g.lpart = reinterpret_cast<float *>(static_cast<char *>(d_workspace) + (((size_t)nb * g.n_params * sizeof(float) + 255) & ~(size_t)255));
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(ddpg_wide_grad_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(ddpg_wide_grad_kernel)");
        if (rc) return rc;
    }
    ApplyArgs ap{};
    ap.d = *d; ap.gpart = g.gpart; ap.lpart = g.lpart; ap.n_blocks = nb; ap.nA = nA; ap.nC = nC;
    const unsigned apply_blocks = (unsigned)((nA + nC + 255) / 256);
    for (int it = 0; it < n_iters; ++it) {
        g.batch_idx = d_batch_idx + (int64_t)it * d->batch_size;
        hipLaunchKernelGGL(ddpg_wide_grad_kernel, dim3(nb), dim3(kWThreads), lds, stream, g);
        ap.it = it;
        ap.losses = d_losses ? d_losses + 2 * it : nullptr;
        hipLaunchKernelGGL(ddpg_wide_apply_kernel, dim3(apply_blocks), dim3(256), 0, stream, ap);
    }
    hipLaunchKernelGGL(ddpg_wide_finish_kernel, dim3(1), dim3(1), 0, stream, d->adam_t, n_iters);
    return check_launch("ssc_ddpg_train (wide)");
}

}  // namespace ssc
