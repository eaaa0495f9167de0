// ddpg_train_fixed.hip -- the DDPG learner step (DDPG_editted.train() + update_target_net(),
// DDPG_Baselines_editted/ddpg_editted.py:287-339; graph :127-133,168-199) compiled for the shape every shipped run
// uses: batch 64, actor obs-64-32-1, critic obs-64-(+a)-32-1, a 1-d action.  Other shapes go through the step
// interpreter of ddpg_train.hip; both are checked against the same fp64 oracle.
//
// Why a second kernel: in the interpreter an iteration is 29 generic steps, and per-step cost is dominated by what
// is NOT arithmetic -- descriptor decode (~500 cycles at one instruction per ~4 cycles per wave), a block barrier,
// the LDS round trip in front of the first FMA.  With the shape fixed at compile time
//   * every LDS address is a constant, every loop is unrolled, and independent layers share a barrier interval:
//     the dependency graph of one iteration has TEN levels (below), not 29;
//   * all Adam moments live in REGISTERS for the whole launch (each element has one owner lane), so the optimiser
//     touches no memory but the parameter images in LDS;
//   * the target networks are updated in the same breath as the online ones (same owner lane), not in a pass of
//     their own;
//   * barriers are `s_waitcnt lgkmcnt(0); s_barrier` -- only LDS data is exchanged between waves, so the global
//     prefetch of the NEXT batch (rows and, one iteration earlier, its indices) stays in flight across them
//     (__syncthreads() drains vmcnt every time).
// The wide contractions (64 -> 32 forward, its transpose, its weight gradient) run on v_mfma_f32_16x16x4_f32: exact
// fp32 products and sums.  One CU sustains 256 fp32 MFMA flop/cycle, so each 64 x 64 x 32 contraction is >= 1024
// cycles; an iteration holds nine of them (three forward passes of the targets / online nets + two critic layers on
// (s, pi(s)), two backward, two weight gradients): ~9.2 k cycles of matrix work, the floor of this design.
//
// Levels of one iteration (B = barrier):
//   L0  batch rows (prefetched) -> S, S2, action row; MpiAdam step sizes                                          B
//   L1  layer 1 of all four nets (target actor / critic on s2, critic / actor on s)                               B
//   L2  layer 2: target actor, actor, critic(s, a)                      [3 MFMA tiles per wave]                  B
//   L3  output layers: target action, pi(s), Q(s, a)                    [one wave each]                           B
//   L4  layer 2: target critic(s2, pi'(s2)), critic(s, pi(s))          [2 MFMA tiles per wave]                  B
//   L5  Q' -> y -> critic loss, dQ; Q(s, pi) -> actor loss; delta of critic layer 2 on the (s, pi) path           B
//   L6  delta of critic layer 2 (TD path); d(-Q)/d(action) -> delta of the actor output                           B
//   L7  critic layer-1 delta [MFMA]; actor layer-2 delta                                                           B
//   L8  actor layer-1 delta [MFMA]; critic gradients + Adam + target update [MFMA + one small element per lane]   B
//   L9  actor gradients + Adam + target update [MFMA + small]; losses out                                         B
#include "ddpg_device.h"
#include "ssc_host.h"

namespace ssc {
namespace {

constexpr int kT = 512, kNW = kT / 64;
constexpr int H1 = 64, H2 = 32;
constexpr int W2S = H2 + 1;   // LDS row stride of a W2 matrix: the backward pass reads COLUMN slices of it (lane = input
                              // row), which at stride 32 put all 16 lanes of a k group on one bank; 33 leaves 2-way
constexpr int al4(int x) { return (x + 3) & ~3; }

// LDS image / flat layout of one net.  IN2 = rows of W2 (H1 for the actor, H1 + 1 for the critic: the action row).
template <int O, int IN2>
struct Net {
    static constexpr int W1 = 0, b1 = O * H1, W2 = b1 + H1, b2 = al4(W2 + IN2 * W2S), W3 = b2 + H2, b3 = W3 + H2, size = al4(b3 + 1);
    static constexpr int gW1 = 0, gb1 = O * H1, gW2 = gb1 + H1, gb2 = gW2 + IN2 * H2, gW3 = gb2 + H2, gb3 = gW3 + H2, gsize = gb3 + 1;
    __device__ static int to_lds(int e) {   // flat index -> image index
        if (e < gW2) return e;
        if (e < gb2) { const int k = e - gW2; return W2 + (k >> 5) * W2S + (k & 31); }
        return b2 + (e - gb2);
    }
};

// activation rows ([unit][kP] floats), in rows
template <int O>
struct Rows {
    static constexpr int S = 0, S2 = O;
    static constexpr int T1 = 2 * O;            // target actor layer 1;  later dzb2 (rows 0..31, L5) and the actor layer-1 delta (L8)
    static constexpr int X2B = T1 + H1;         // target critic layer 1 + target action row;  later the critic layer-1 delta (L7)
    static constexpr int X2 = X2B + H1 + 1;     // critic layer 1 + the batch action row
    static constexpr int U1 = X2 + H1 + 1;      // actor layer 1
    static constexpr int T2 = U1 + H1;          // target actor layer 2;  later the actor layer-2 delta (L7)
    static constexpr int U2 = T2 + H2;          // actor layer 2
    static constexpr int CA2 = U2 + H2;         // critic layer 2 on (s, a)
    static constexpr int TB2 = CA2 + H2;        // target critic layer 2;  later the critic layer-2 delta (L6)
    static constexpr int CB2 = TB2 + H2;        // critic layer 2 on (s, pi(s))
    static constexpr int PI = CB2 + H2, Q = PI + 1, DQ = Q + 1, DZ3A = DQ + 1;
    static constexpr int total = DZ3A + 1;
    static constexpr int DZB2 = T1, DZ1A = T1, DZ1 = X2B, DZ2A = T2, DZ2 = TB2;
};

struct FixedArgs {
    ssc_ddpg_desc d;
    ssc_replay_view rp;
    const int32_t *batch_idx;
    float *losses;
    int32_t n_iters;
};

// only LDS data crosses waves: no vmcnt drain (see the header)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool TANH2> __device__ __forceinline__ float act2(float v) { return TANH2 ? tanh_fast(v) : fmaxf(v, 0.0f); }
template <bool TANH2> __device__ __forceinline__ float act2_deriv(float a) { return TANH2 ? 1.0f - a * a : (a > 0.0f ? 1.0f : 0.0f); }

// one 16 x 16 tile of Z = bias + W2^T X over the 64 head rows;  rows = units j0.., cols = samples b0..
__device__ __forceinline__ f32x4m fwd_tile(const float *W2, const float *b2, const float *X, int j0, int b0, int c, int kg) {
    f32x4m acc = *reinterpret_cast<const f32x4m *>(b2 + j0 + 4 * kg);
    const float *wp = W2 + kg * W2S + j0 + c, *xp = X + kg * kP + b0 + c;
    float a[H1 / 4], b[H1 / 4];
#pragma unroll
    for (int s = 0; s < H1 / 4; ++s) { a[s] = wp[4 * s * W2S]; b[s] = xp[4 * s * kP]; }
#pragma unroll
    for (int s = 0; s < H1 / 4; ++s) acc = mfma4(a[s], b[s], acc);
    return acc;
}

template <bool TANH2>
__device__ __forceinline__ void store_tile(float *Z, const f32x4m &acc, int j0, int b0, int c, int kg) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Z[(j0 + 4 * kg + r) * kP + b0 + c] = act2<TANH2>(acc[r]);
}

// dX[i][b] = relu'(A[i][b]) sum_j W2[i][j] dZ[j][b] for input rows 16 ib.. and the sample tiles 2 sp, 2 sp + 1
__device__ __forceinline__ void bwd_item(const float *W2, const float *dZ, const float *A, float *dX, int ib, int sp, int c, int kg) {
    const float *wp = W2 + (16 * ib + c) * W2S + kg;
    const float *z0 = dZ + kg * kP + 32 * sp + c;
    float a[H2 / 4], u0[H2 / 4], u1[H2 / 4];
#pragma unroll
    for (int s = 0; s < H2 / 4; ++s) { a[s] = wp[4 * s]; u0[s] = z0[4 * s * kP]; u1[s] = z0[4 * s * kP + 16]; }
    f32x4m acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
#pragma unroll
    for (int s = 0; s < H2 / 4; ++s) { acc0 = mfma4(a[s], u0[s], acc0); acc1 = mfma4(a[s], u1[s], acc1); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = (16 * ib + 4 * kg + r) * kP + 32 * sp + c;
        const float a0 = A[o], a1 = A[o + 16];          // read BEFORE the write: dX may alias nothing A uses, but keep the order plain
        dX[o] = a0 > 0.0f ? acc0[r] : 0.0f;
        dX[o + 16] = a1 > 0.0f ? acc1[r] : 0.0f;
    }
}

// dW[i][j] = sum_b X[i][b] dZ[j][b] for rows 16 ib.., units 16 jb..;  lane holds (i = 16 ib + 4 kg + r, j = 16 jb + c)
__device__ __forceinline__ f32x4m wgrad_tile(const float *X, const float *dZ, int ib, int jb, int c, int kg) {
    const float *xp = X + (16 * ib + c) * kP + kg, *zp = dZ + (16 * jb + c) * kP + kg;
    float a[kB / 4], b[kB / 4];
#pragma unroll
    for (int s = 0; s < kB / 4; ++s) { a[s] = xp[4 * s]; b[s] = zp[4 * s]; }
    f32x4m acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < kB / 4; ++s) acc = mfma4(a[s], b[s], acc);
    return acc;
}

// out = bias + sum_i W3[i] X[i][b], 32 inputs, lane = sample
__device__ __forceinline__ float dot32(const float *W3, const float *b3, const float *X, int b) {
    float acc0 = b3[0], acc1 = 0.0f;
#pragma unroll
    for (int q = 0; q < H2 / 4; ++q) {
        const f4 w = *reinterpret_cast<const f4 *>(W3 + 4 * q);
        acc0 = fmaf(w[0], X[(4 * q) * kP + b], acc0);
        acc1 = fmaf(w[1], X[(4 * q + 1) * kP + b], acc1);
        acc0 = fmaf(w[2], X[(4 * q + 2) * kP + b], acc0);
        acc1 = fmaf(w[3], X[(4 * q + 3) * kP + b], acc1);
    }
    return acc0 + acc1;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// MpiAdam.update on one element (baselines common/mpi_adam.py [third-party], ddpg_editted.py:326-327), then
// update_target_net on the same element (:338-339).  theta / target in LDS, moments in registers.
__device__ __forceinline__ void adam_target(float *theta, float *target, int idx, float &m, float &v, float g, const AdamCfg &c, float tau) {
    m = c.beta1 * m + (1.0f - c.beta1) * g;
    v = c.beta2 * v + (1.0f - c.beta2) * (g * g);
    const float th = theta[idx] + (-c.a) * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + c.eps);
    theta[idx] = th;
    target[idx] = (1.0f - tau) * target[idx] + tau * th;
}

// One "small" parameter element: its gradient is a 64-sample dot product of two LDS rows (or the sum of one).
struct SmallElem {
    int x, z;        // LDS float offsets of the input row (-1: bias) and the delta row
    int lidx, gidx;  // image / flat index of the parameter
    bool on;
};

__device__ __forceinline__ float small_grad(const float *lds, const SmallElem &e) {
    f4 g4 = (f4)(0.0f);
    if (e.x >= 0) {
#pragma unroll
        for (int q = 0; q < kB / 4; ++q)
            g4 += *reinterpret_cast<const f4 *>(lds + e.x + 4 * q) * *reinterpret_cast<const f4 *>(lds + e.z + 4 * q);
    } else {
#pragma unroll
        for (int q = 0; q < kB / 4; ++q) g4 += *reinterpret_cast<const f4 *>(lds + e.z + 4 * q);
    }
    return (g4[0] + g4[1]) + (g4[2] + g4[3]);
}

template <int O, bool TANH2>
__global__ __launch_bounds__(kT) void ddpg_train_fixed_kernel(FixedArgs g) {
    using NA = Net<O, H1>;
    using NC = Net<O, H1 + 1>;
    using R = Rows<O>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[2];
    __shared__ AdamCfg cfg_s[2];
    const ssc_ddpg_desc &d = g.d;
    const int tid = threadIdx.x, wave = tid >> 6, b = tid & 63, c = tid & 15, kg = (tid >> 4) & 3;
    float *const act = lds;                                   // activation rows
    float *const th_a = lds + R::total * kP;                  // parameter images
    float *const th_c = th_a + NA::size;
    float *const th_ta = th_c + NC::size;
    float *const th_tc = th_ta + NA::size;
    auto row = [&](int r) { return act + r * kP; };

    for (int e = tid; e < NA::gsize; e += kT) { const int l = NA::to_lds(e); th_a[l] = d.actor[e]; th_ta[l] = d.target_actor[e]; }
    for (int e = tid; e < NC::gsize; e += kT) { const int l = NC::to_lds(e); th_c[l] = d.critic[e]; th_tc[l] = d.target_critic[e]; }

    // ---- owners of the Adam moments ------------------------------------------------------------------------------
    // wide tiles (W2 rows 0..63 of both nets): wave -> (ib = wave >> 1, jb = wave & 1), lane -> 4 elements
    const int w_ib = wave >> 1, w_jb = wave & 1;
    float mW2c[4], vW2c[4], mW2a[4], vW2a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = (16 * w_ib + 4 * kg + r) * H2 + 16 * w_jb + c;
        mW2c[r] = d.adam_m_critic[NC::gW2 + k]; vW2c[r] = d.adam_v_critic[NC::gW2 + k];
        mW2a[r] = d.adam_m_actor[NA::gW2 + k];  vW2a[r] = d.adam_v_actor[NA::gW2 + k];
    }
    // small elements: lane tid owns critic element tid and actor element tid of the lists below
    SmallElem ec, ea;
    {
        int e = tid;   // critic: W1 | b1 | W2 row 64 (the action row) | b2 | W3 | b3
        ec.on = true;
        if (e < O * H1) { ec.x = (R::S + e / H1) * kP; ec.z = (R::DZ1 + e % H1) * kP; ec.gidx = NC::gW1 + e; }
        else if ((e -= O * H1) < H1) { ec.x = -1; ec.z = (R::DZ1 + e) * kP; ec.gidx = NC::gb1 + e; }
        else if ((e -= H1) < H2) { ec.x = (R::X2 + H1) * kP; ec.z = (R::DZ2 + e) * kP; ec.gidx = NC::gW2 + H1 * H2 + e; }
        else if ((e -= H2) < H2) { ec.x = -1; ec.z = (R::DZ2 + e) * kP; ec.gidx = NC::gb2 + e; }
        else if ((e -= H2) < H2) { ec.x = (R::CA2 + e) * kP; ec.z = R::DQ * kP; ec.gidx = NC::gW3 + e; }
        else if ((e -= H2) < 1) { ec.x = -1; ec.z = R::DQ * kP; ec.gidx = NC::gb3; }
        else { ec.on = false; ec.x = -1; ec.z = 0; ec.gidx = 0; }
        ec.lidx = NC::to_lds(ec.gidx);
        e = tid;       // actor: W1 | b1 | b2 | W3 | b3
        ea.on = true;
        if (e < O * H1) { ea.x = (R::S + e / H1) * kP; ea.z = (R::DZ1A + e % H1) * kP; ea.gidx = NA::gW1 + e; }
        else if ((e -= O * H1) < H1) { ea.x = -1; ea.z = (R::DZ1A + e) * kP; ea.gidx = NA::gb1 + e; }
        else if ((e -= H1) < H2) { ea.x = -1; ea.z = (R::DZ2A + e) * kP; ea.gidx = NA::gb2 + e; }
        else if ((e -= H2) < H2) { ea.x = (R::U2 + e) * kP; ea.z = R::DZ3A * kP; ea.gidx = NA::gW3 + e; }
        else if ((e -= H2) < 1) { ea.x = -1; ea.z = R::DZ3A * kP; ea.gidx = NA::gb3; }
        else { ea.on = false; ea.x = -1; ea.z = 0; ea.gidx = 0; }
        ea.lidx = NA::to_lds(ea.gidx);
    }
    static_assert(O * H1 + H1 + 3 * H2 + 1 <= kT, "one small critic element per lane");
    float mc = 0.0f, vc = 0.0f, ma = 0.0f, va = 0.0f;
    if (ec.on) { mc = d.adam_m_critic[ec.gidx]; vc = d.adam_v_critic[ec.gidx]; }
    if (ea.on) { ma = d.adam_m_actor[ea.gidx]; va = d.adam_v_actor[ea.gidx]; }

    int tA = d.adam_t[0], tC = d.adam_t[1];
    // running beta powers for MpiAdam's bias correction, in f64 (1 - 0.999^t loses 5 digits in fp32)
    double b1a = ipow((double)d.beta1, tA), b2a = ipow((double)d.beta2, tA);
    double b1c = ipow((double)d.beta1, tC), b2c = ipow((double)d.beta2, tC);

    // ---- batch pipeline: indices two iterations ahead, rows one iteration ahead (wave 0, lane = sample) ----------
    float pf_s[O], pf_s2[O], pf_a = 0.0f, pf_r = 0.0f, pf_t = 0.0f;
    int64_t rec_next = 0;
    auto fetch_rows = [&](int64_t rec) {
#pragma unroll
        for (int k = 0; k < O; ++k) { pf_s[k] = g.rp.s[rec * O + k]; pf_s2[k] = g.rp.s2[rec * O + k]; }
        pf_a = g.rp.a[rec];
        pf_r = g.rp.r[rec];
        pf_t = g.rp.t[rec] ? 1.0f : 0.0f;
    };
    auto idx_of = [&](int it_) { return (int64_t)g.batch_idx[(int64_t)(it_ < g.n_iters ? it_ : g.n_iters - 1) * kB + tid]; };
    if (tid < kB) {
        fetch_rows(idx_of(0));
        rec_next = idx_of(1);
    }
    __syncthreads();   // parameter images complete

    float r_cur = 0.0f, t_cur = 0.0f;
    for (int it = 0; it < g.n_iters; ++it) {
        // ---- L0: ReplayBuffer.sample_batch rows; MpiAdam step sizes --------------------------------------------------
        ++tA; ++tC;
        b1a *= (double)d.beta1; b2a *= (double)d.beta2; b1c *= (double)d.beta1; b2c *= (double)d.beta2;
        if (tid == kB) {
            cfg_s[0] = AdamCfg{(float)((double)d.actor_lr * sqrt(1.0 - b2a) / (1.0 - b1a)), d.beta1, d.beta2, d.epsilon};
            cfg_s[1] = AdamCfg{(float)((double)d.critic_lr * sqrt(1.0 - b2c) / (1.0 - b1c)), d.beta1, d.beta2, d.epsilon};
        }
        if (tid < kB) {
#pragma unroll
            for (int k = 0; k < O; ++k) {   // obs0 / obs1 enter every network clipped (ddpg_editted.py:106-109)
                row(R::S + k)[tid] = d.obs_clip > 0.0f ? fminf(fmaxf(pf_s[k], -d.obs_clip), d.obs_clip) : pf_s[k];
                row(R::S2 + k)[tid] = d.obs_clip > 0.0f ? fminf(fmaxf(pf_s2[k], -d.obs_clip), d.obs_clip) : pf_s2[k];
            }
            row(R::X2 + H1)[tid] = pf_a;
            r_cur = pf_r;
            t_cur = pf_t;
            fetch_rows(rec_next);        // rows of iteration it + 1: in flight until the next L0
            rec_next = idx_of(it + 2);
        }
        lds_barrier();

        // ---- L1: layer 1 of the four nets; wave -> units 8 wave .., lane = sample ----------------------------------
        {
            float xs[O], x2[O];
#pragma unroll
            for (int k = 0; k < O; ++k) { xs[k] = row(R::S + k)[b]; x2[k] = row(R::S2 + k)[b]; }
            auto layer1 = [&](const float *img, const float *x, float *Z) {   // W1 / b1 sit at the same offsets in both nets
                f4 a0 = *reinterpret_cast<const f4 *>(img + NA::b1 + 8 * wave), a1 = *reinterpret_cast<const f4 *>(img + NA::b1 + 8 * wave + 4);
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    a0 += *reinterpret_cast<const f4 *>(img + NA::W1 + k * H1 + 8 * wave) * x[k];
                    a1 += *reinterpret_cast<const f4 *>(img + NA::W1 + k * H1 + 8 * wave + 4) * x[k];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Z[(8 * wave + e) * kP + b] = fmaxf(a0[e], 0.0f);
                    Z[(8 * wave + 4 + e) * kP + b] = fmaxf(a1[e], 0.0f);
                }
            };
            layer1(th_ta, x2, row(R::T1));
            layer1(th_tc, x2, row(R::X2B));
            layer1(th_c, xs, row(R::X2));
            layer1(th_a, xs, row(R::U1));
        }
        lds_barrier();

        // ---- L2: layer 2 of target actor, actor, critic(s, a); wave -> tile (units 16 (wave >> 2).., samples 16 (wave & 3)..)
        {
            const int j0 = 16 * (wave >> 2), b0 = 16 * (wave & 3);
            f32x4m t2 = fwd_tile(th_ta + NA::W2, th_ta + NA::b2, row(R::T1), j0, b0, c, kg);
            f32x4m u2 = fwd_tile(th_a + NA::W2, th_a + NA::b2, row(R::U1), j0, b0, c, kg);
            f32x4m c2 = fwd_tile(th_c + NC::W2, th_c + NC::b2, row(R::X2), j0, b0, c, kg);
            c2 += *reinterpret_cast<const f32x4m *>(th_c + NC::W2 + H1 * W2S + j0 + 4 * kg) * row(R::X2 + H1)[b0 + c];   // the action row
            store_tile<TANH2>(row(R::T2), t2, j0, b0, c, kg);
            store_tile<TANH2>(row(R::U2), u2, j0, b0, c, kg);
            store_tile<TANH2>(row(R::CA2), c2, j0, b0, c, kg);
        }
        lds_barrier();

        // ---- L3: the three output layers, one wave each ------------------------------------------------------------
        if (wave == 0) row(R::X2B + H1)[b] = tanh_fast(dot32(th_ta + NA::W3, th_ta + NA::b3, row(R::T2), b));   // pi'(s2)  (:132)
        else if (wave == 1) row(R::PI)[b] = tanh_fast(dot32(th_a + NA::W3, th_a + NA::b3, row(R::U2), b));       // pi(s)    (:127)
        else if (wave == 2) row(R::Q)[b] = dot32(th_c + NC::W3, th_c + NC::b3, row(R::CA2), b);                  // Q(s, a)  (:181)
        lds_barrier();

        // ---- L4: layer 2 of target critic(s2, pi'(s2)) and critic(s, pi(s)) ----------------------------------------
        {
            const int j0 = 16 * (wave >> 2), b0 = 16 * (wave & 3);
            f32x4m tb = fwd_tile(th_tc + NC::W2, th_tc + NC::b2, row(R::X2B), j0, b0, c, kg);
            f32x4m cb = fwd_tile(th_c + NC::W2, th_c + NC::b2, row(R::X2), j0, b0, c, kg);
            tb += *reinterpret_cast<const f32x4m *>(th_tc + NC::W2 + H1 * W2S + j0 + 4 * kg) * row(R::X2B + H1)[b0 + c];
            cb += *reinterpret_cast<const f32x4m *>(th_c + NC::W2 + H1 * W2S + j0 + 4 * kg) * row(R::PI)[b0 + c];
            store_tile<TANH2>(row(R::TB2), tb, j0, b0, c, kg);
            store_tile<TANH2>(row(R::CB2), cb, j0, b0, c, kg);
        }
        lds_barrier();

        // ---- L5 -------------------------------------------------------------------------------------------------------
        if (wave == 0) {          // target_Q = r + (1 - terminal) gamma Q'(s2, pi'(s2))  (:132-133); critic loss = mean((Q - y)^2)  (:181)
            const float qt = dot32(th_tc + NC::W3, th_tc + NC::b3, row(R::TB2), b);
            const float y = r_cur + (1.0f - t_cur) * d.gamma * qt;
            const float e = row(R::Q)[b] - y;
            row(R::DQ)[b] = 2.0f * e / (float)kB;
            const float s = wave_sum(e * e);
            if (b == 0) red[0] = s;
        } else if (wave == 1) {   // actor loss = -mean Q(s, pi(s))  (:168)
            const float s = wave_sum(-dot32(th_c + NC::W3, th_c + NC::b3, row(R::CB2), b));
            if (b == 0) red[1] = s;
        } else {                  // (s, pi) path: dq = -1/B, delta of critic layer 2 = W3 dq act2'(.)
            for (int e = tid - 2 * 64; e < H2 * kB; e += kT - 2 * 64) {
                const int o = (e >> 6) * kP + (e & 63);
                row(R::DZB2)[o] = th_c[NC::W3 + (e >> 6)] * (-1.0f / (float)kB) * act2_deriv<TANH2>(row(R::CB2)[o]);
            }
        }
        lds_barrier();

        // ---- L6: TD path delta of critic layer 2; d(-Q)/d(action) through the output tanh ---------------------------
#pragma unroll
        for (int q = 0; q < H2 * kB / kT; ++q) {
            const int e = tid + kT * q, o = (e >> 6) * kP + (e & 63);
            row(R::DZ2)[o] = th_c[NC::W3 + (e >> 6)] * row(R::DQ)[e & 63] * act2_deriv<TANH2>(row(R::CA2)[o]);
        }
        if (wave == kNW - 1) {
            float da0 = 0.0f, da1 = 0.0f;
#pragma unroll
            for (int j = 0; j < H2; j += 2) {
                da0 = fmaf(th_c[NC::W2 + H1 * W2S + j], row(R::DZB2 + j)[b], da0);
                da1 = fmaf(th_c[NC::W2 + H1 * W2S + j + 1], row(R::DZB2 + j + 1)[b], da1);
            }
            const float p = row(R::PI)[b];
            row(R::DZ3A)[b] = (da0 + da1) * (1.0f - p * p);
        }
        lds_barrier();

        // ---- L7: critic layer-1 delta (MFMA: wave -> 16 input rows x 32 samples); actor layer-2 delta ------------------
        bwd_item(th_c + NC::W2, row(R::DZ2), row(R::X2), row(R::DZ1), wave >> 1, wave & 1, c, kg);
#pragma unroll
        for (int q = 0; q < H2 * kB / kT; ++q) {
            const int e = tid + kT * q, o = (e >> 6) * kP + (e & 63);
            row(R::DZ2A)[o] = th_a[NA::W3 + (e >> 6)] * row(R::DZ3A)[e & 63] * act2_deriv<TANH2>(row(R::U2)[o]);
        }
        lds_barrier();

        // ---- L8: actor layer-1 delta; critic gradients -> MpiAdam -> target critic -----------------------------------
        bwd_item(th_a + NA::W2, row(R::DZ2A), row(R::U1), row(R::DZ1A), wave >> 1, wave & 1, c, kg);
        {
            const AdamCfg cc = cfg_s[1];
            const f32x4m gw = wgrad_tile(row(R::X2), row(R::DZ2), w_ib, w_jb, c, kg);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                adam_target(th_c, th_tc, NC::W2 + (16 * w_ib + 4 * kg + r) * W2S + 16 * w_jb + c, mW2c[r], vW2c[r], gw[r], cc, d.tau);
            if (ec.on) adam_target(th_c, th_tc, ec.lidx, mc, vc, small_grad(lds, ec), cc, d.tau);
        }
        lds_barrier();

        // ---- L9: actor gradients -> MpiAdam -> target actor; losses ------------------------------------------------------
        {
            const AdamCfg ca = cfg_s[0];
            const f32x4m gw = wgrad_tile(row(R::U1), row(R::DZ2A), w_ib, w_jb, c, kg);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                adam_target(th_a, th_ta, NA::W2 + (16 * w_ib + 4 * kg + r) * W2S + 16 * w_jb + c, mW2a[r], vW2a[r], gw[r], ca, d.tau);
            if (ea.on) adam_target(th_a, th_ta, ea.lidx, ma, va, small_grad(lds, ea), ca, d.tau);
        }
        if (tid == 0 && g.losses != nullptr) {
            g.losses[2 * it + 0] = red[0] / (float)kB;
            g.losses[2 * it + 1] = red[1] / (float)kB;
        }
        lds_barrier();
    }

    // the forward kernels outside (ssc_actor_forward, ssc_critic_forward, rollouts) read the global arrays
    for (int e = tid; e < NA::gsize; e += kT) { const int l = NA::to_lds(e); d.actor[e] = th_a[l]; d.target_actor[e] = th_ta[l]; }
    for (int e = tid; e < NC::gsize; e += kT) { const int l = NC::to_lds(e); d.critic[e] = th_c[l]; d.target_critic[e] = th_tc[l]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = (16 * w_ib + 4 * kg + r) * H2 + 16 * w_jb + c;
        d.adam_m_critic[NC::gW2 + k] = mW2c[r]; d.adam_v_critic[NC::gW2 + k] = vW2c[r];
        d.adam_m_actor[NA::gW2 + k] = mW2a[r];  d.adam_v_actor[NA::gW2 + k] = vW2a[r];
    }
    if (ec.on) { d.adam_m_critic[ec.gidx] = mc; d.adam_v_critic[ec.gidx] = vc; }
    if (ea.on) { d.adam_m_actor[ea.gidx] = ma; d.adam_v_actor[ea.gidx] = va; }
    if (tid == 0) { d.adam_t[0] = tA; d.adam_t[1] = tC; }
}

template <int O, bool TANH2>
int launch_fixed(const FixedArgs &g, hipStream_t stream) {
    const size_t lds = ((size_t)Rows<O>::total * kP + 2 * (Net<O, H1>::size + Net<O, H1 + 1>::size)) * sizeof(float);
    static_assert(((size_t)Rows<O>::total * kP + 2 * (Net<O, H1>::size + Net<O, H1 + 1>::size)) * sizeof(float) <= 160 * 1024 - 256,
                  "activation rows + parameter images must fit the LDS");
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(ddpg_train_fixed_kernel<O, TANH2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "hipFuncSetAttribute(ddpg_train_fixed_kernel)");
    if (rc) return rc;
    hipLaunchKernelGGL((ddpg_train_fixed_kernel<O, TANH2>), dim3(1), dim3(kT), lds, stream, g);
    return check_launch("ssc_ddpg_train");
}

}  // namespace

bool ddpg_fixed_shape(const ssc_ddpg_desc *d) {
    return d->batch_size == kB && (d->obs_dim == 2 || d->obs_dim == 3) && d->act_dim == 1 && d->actor_h1 == H1 && d->actor_h2 == H2 &&
           d->critic_h1 == H1 && d->critic_h2 == H2;
}

// arguments already validated by ssc_ddpg_train
int ddpg_train_fixed(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                     float *d_losses, hipStream_t stream) {
    FixedArgs g{*d, *rp, d_batch_idx, d_losses, n_iters};
    const bool t2 = d->last_layer_tanh != 0;
    if (d->obs_dim == 2) return t2 ? launch_fixed<2, true>(g, stream) : launch_fixed<2, false>(g, stream);
    return t2 ? launch_fixed<3, true>(g, stream) : launch_fixed<3, false>(g, stream);
}

}  // namespace ssc
