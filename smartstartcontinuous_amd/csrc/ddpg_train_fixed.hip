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
// cycles; an iteration holds eight of them (layer 2 of the four nets -- critic(s, pi(s)) reuses the head of
// critic(s, a) --, two backward, two weight gradients): ~8.2 k cycles of matrix work, the floor of this design.
//
// Levels of one iteration (B = barrier):
//   L0  batch rows (prefetched) -> S, S2, action row; MpiAdam step sizes                                          B
//   L1  layer 1 of all four nets (target actor / critic on s2, critic / actor on s)  [one MFMA k-step per tile]   B
//   L2  layer 2: target actor, actor, critic(s, a) + the head of target critic layer 2  [4 MFMA tiles per wave]  B
//   L3  output layers: target action, pi(s), Q(s, a)                    [one wave each]                           B
//   L4  layer 2: target critic(s2, pi'(s2)), critic(s, pi(s)): action rows onto the heads kept from L2          B
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
constexpr int W2S = H2 + 4;   // LDS row stride of a W2 matrix.  The backward pass reads COLUMN slices of it (lane = input row): at
                              // stride 32 all 16 lanes of a k group sit on one bank, at 36 they spread over 8 (2-way).  The forward
                              // pass wants rows 4 apart to sit 16 banks apart (see fwd_item): 36 = 4 mod 32 does that too, like kP
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
    static constexpr int LC = DZ3A + 1, LA = LC + 1;   // per-sample loss terms (summed one level later, off the critical path)
    static constexpr int total = LA + 1;
    static constexpr int DZB2 = T1, DZ1A = T1, DZ1 = X2B, DZ2A = T2, DZ2 = TB2;
};

struct FixedArgs {
    ssc_ddpg_desc d;
    ssc_replay_view rp;
    const int32_t *batch_idx;
    float *losses;
    int32_t n_iters;
    // TILED (a batch of several 64-row tiles, one workgroup each, ONE iteration per launch; ddpg_train_fixed_tiled):
    float inv_b;        // 1 / batch_size
    float *gpart;       // [tiles][actor gsize + critic gsize]: this tile's share of every gradient
    float *lpart;       // [tiles][2]: its sums of (Q - y)^2 and -Q(s, pi(s))
};

// Diagnostic build (-DSSC_DDPG_DIAG, tools/exp_ddpg_phases.py): d_losses is [n_iters][16] and receives the cycles
// thread 0 spent in each level instead of the losses.
#ifdef SSC_DDPG_DIAG
#define LEVEL_MARK(k) do { __builtin_amdgcn_sched_barrier(0); const uint64_t now_ = __builtin_amdgcn_s_memtime(); \
        if (tid == 0) g.losses[16 * it + (k)] = (float)(now_ - cp_prev); cp_prev = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LEVEL_MARK(k) do { } while (0)
#endif

// only LDS data crosses waves: no vmcnt drain (see the header)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool TANH2> __device__ __forceinline__ float act2(float v) { return TANH2 ? tanh_fast(v) : fmaxf(v, 0.0f); }
template <bool TANH2> __device__ __forceinline__ float act2_deriv(float a) { return TANH2 ? 1.0f - a * a : (a > 0.0f ? 1.0f : 0.0f); }

// Two 16 x 16 tiles of Z = bias + W2^T X over the 64 head rows: units 16 jb.., samples 32 sp.. and 32 sp + 16..; the W
// fragment is fetched once for both.  k-step s of lane group kg contracts input row k(s, kg) = 16 (s >> 2) + (s & 3)
// + 4 kg instead of the usual 4 s + kg: a ds_read_b32 is banked (address / 4) % 32 over each 32-lane half, i.e. over the
// lane groups kg and kg + 1 together; both operands have row strides = 4 mod 32 (kP = 68, W2S = 36), so rows FOUR apart
// land 16 banks apart and every read is conflict-free (rows one apart overlap in 12 of 16 banks: 2-way).  The LDS, not
// the matrix pipe, bounded these levels: 32 operand reads per 16 MFMAs at ~4 cycles each vs 32 cycles per MFMA / 4 SIMDs.
struct TilePair { f32x4m t0, t1; };
__device__ __forceinline__ TilePair fwd_item(const float *W2, const float *b2, const float *X, int jb, int sp, int c, int kg) {
    TilePair acc;
    acc.t0 = acc.t1 = *reinterpret_cast<const f32x4m *>(b2 + 16 * jb + 4 * kg);
    const float *wp = W2 + 4 * kg * W2S + 16 * jb + c, *xp = X + 4 * kg * kP + 32 * sp + c;
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // two batches of 8 k-steps: 24 operand registers in flight, not 48
        float a[H1 / 8], u0[H1 / 8], u1[H1 / 8];
#pragma unroll
        for (int s = 0; s < H1 / 8; ++s) {
            const int k = 32 * h + 16 * (s >> 2) + (s & 3);
            a[s] = wp[k * W2S]; u0[s] = xp[k * kP]; u1[s] = xp[k * kP + 16];
        }
#pragma unroll
        for (int s = 0; s < H1 / 8; ++s) { acc.t0 = mfma4(a[s], u0[s], acc.t0); acc.t1 = mfma4(a[s], u1[s], acc.t1); }
    }
    return acc;
}

// + the 65th input row (critic: the action), activation, store
template <bool TANH2>
__device__ __forceinline__ void store_item(float *Z, TilePair acc, const float *w_tail, const float *x_tail, int jb, int sp, int c, int kg) {
    if (w_tail != nullptr) {
        const f32x4m w = *reinterpret_cast<const f32x4m *>(w_tail + 16 * jb + 4 * kg);
        acc.t0 += w * x_tail[32 * sp + c];
        acc.t1 += w * x_tail[32 * sp + 16 + c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float *z = Z + (16 * jb + 4 * kg + r) * kP + 32 * sp + c;
        z[0] = act2<TANH2>(acc.t0[r]);
        z[16] = act2<TANH2>(acc.t1[r]);
    }
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

// dW[i][j] = sum_b X[i][b] dZ[j][b] for rows 16 ib.., units 16 jb..;  lane holds (i = 16 ib + 4 kg + r, j = 16 jb + c).
// The contraction index (the sample) is the contiguous one in both operands: lane group kg takes samples 16 q + 4 kg ..
// + 3 as FOUR k-steps from one ds_read_b128 per operand -- 8 reads per tile instead of 32.
__device__ __forceinline__ f32x4m wgrad_tile(const float *X, const float *dZ, int ib, int jb, int c, int kg) {
    const float *xp = X + (16 * ib + c) * kP + 4 * kg, *zp = dZ + (16 * jb + c) * kP + 4 * kg;
    f4 a[kB / 16], b[kB / 16];
#pragma unroll
    for (int q = 0; q < kB / 16; ++q) { a[q] = *reinterpret_cast<const f4 *>(xp + 16 * q); b[q] = *reinterpret_cast<const f4 *>(zp + 16 * q); }
    f32x4m acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < kB / 16; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma4(a[q][e], b[q][e], acc);
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
// update_target_net on the same element (:338-339).  th / tg: the old values of theta and target (read by the caller,
// all of a lane's elements at once: one LDS round trip, not one per element); moments in registers.
__device__ __forceinline__ void adam_target(float &th, float &tg, float &m, float &v, float g, const AdamCfg &c, float tau) {
    m = c.beta1 * m + (1.0f - c.beta1) * g;
    v = c.beta2 * v + (1.0f - c.beta2) * (g * g);
    th += (-c.a) * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + c.eps);   // 1-ulp sqrt / rcp: a 1e-7 relative wobble
    tg = (1.0f - tau) * tg + tau * th;                                              // of a 1e-3-sized step
}

// the four elements (i = 16 ib + 4 kg + r, j = 16 jb + c) of a weight-gradient tile
__device__ __forceinline__ void adam_tile(float *theta, float *target, int base, const f32x4m &g, float (&m)[4], float (&v)[4],
                                          const AdamCfg &c, float tau) {
    float th[4], tg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { th[r] = theta[base + r * W2S]; tg[r] = target[base + r * W2S]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) adam_target(th[r], tg[r], m[r], v[r], g[r], c, tau);
#pragma unroll
    for (int r = 0; r < 4; ++r) { theta[base + r * W2S] = th[r]; target[base + r * W2S] = tg[r]; }
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

// TILED: the same ten levels on rows [64 blockIdx.x, + 64) of a larger batch, up to the gradients -- which go to the tile's
// slice of `gpart` instead of into MpiAdam (ddpg_wide_apply_kernel sums the slices in tile order and updates).
template <int O, bool TANH2, bool TILED>
__global__ __launch_bounds__(kT) void ddpg_train_fixed_kernel(FixedArgs g) {
    using NA = Net<O, H1>;
    using NC = Net<O, H1 + 1>;
    using R = Rows<O>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[2];
    __shared__ AdamCfg cfg_s[2];
    const ssc_ddpg_desc &d = g.d;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), b = tid & 63, c = tid & 15, kg = (tid >> 4) & 3;
    float *const act = lds;                                   // activation rows
    float *const th_a = lds + R::total * kP;                  // parameter images
    float *const th_c = th_a + NA::size;
    float *const th_ta = th_c + NC::size;
    float *const th_tc = th_ta + NA::size;
    auto row = [&](int r) { return act + r * kP; };

    // (the first batch's record numbers: requested in front of the parameter loads, their round trip runs under them)
    int64_t rec0 = 0;
    if (tid < kB) rec0 = (int64_t)g.batch_idx[(TILED ? (int64_t)blockIdx.x * kB : 0) + tid];
    // The four parameter vectors -> LDS images: ALL loads first (about twenty per thread, indices clamped), then the stores --
    // as a load / store loop this was ten dependent round trips in front of the first level, half of a one-iteration
    // (TILED) launch.
    {
        constexpr int kIA = (NA::gsize + kT - 1) / kT, kIC = (NC::gsize + kT - 1) / kT;
        float va[kIA], vta[kIA], vc[kIC], vtc[kIC];
#pragma unroll
        for (int k = 0; k < kIA; ++k) { const int e = min(tid + k * kT, NA::gsize - 1); va[k] = d.actor[e]; vta[k] = d.target_actor[e]; }
#pragma unroll
        for (int k = 0; k < kIC; ++k) { const int e = min(tid + k * kT, NC::gsize - 1); vc[k] = d.critic[e]; vtc[k] = d.target_critic[e]; }
#pragma unroll
        for (int k = 0; k < kIA; ++k) {
            const int e = tid + k * kT;
            if (e < NA::gsize) { const int l = NA::to_lds(e); th_a[l] = va[k]; th_ta[l] = vta[k]; }
        }
#pragma unroll
        for (int k = 0; k < kIC; ++k) {
            const int e = tid + k * kT;
            if (e < NC::gsize) { const int l = NC::to_lds(e); th_c[l] = vc[k]; th_tc[l] = vtc[k]; }
        }
    }

    // ---- owners of the Adam moments ------------------------------------------------------------------------------
    // wide tiles (W2 rows 0..63 of both nets): wave -> (ib = wave >> 1, jb = wave & 1), lane -> 4 elements
    const int w_ib = wave >> 1, w_jb = wave & 1;
    float mW2c[4] = {}, vW2c[4] = {}, mW2a[4] = {}, vW2a[4] = {};
    if constexpr (!TILED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = (16 * w_ib + 4 * kg + r) * H2 + 16 * w_jb + c;
            mW2c[r] = d.adam_m_critic[NC::gW2 + k]; vW2c[r] = d.adam_v_critic[NC::gW2 + k];
            mW2a[r] = d.adam_m_actor[NA::gW2 + k];  vW2a[r] = d.adam_v_actor[NA::gW2 + k];
        }
    }
    float *const gout = TILED ? g.gpart + (int64_t)blockIdx.x * (NA::gsize + NC::gsize) : nullptr;
    const float inv_b = TILED ? g.inv_b : 1.0f / (float)kB;
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
    if (!TILED && ec.on) { mc = d.adam_m_critic[ec.gidx]; vc = d.adam_v_critic[ec.gidx]; }
    if (!TILED && ea.on) { ma = d.adam_m_actor[ea.gidx]; va = d.adam_v_actor[ea.gidx]; }

    int tA = TILED ? 0 : d.adam_t[0], tC = TILED ? 0 : d.adam_t[1];
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
    auto idx_of = [&](int it_) {
        if constexpr (TILED) return (int64_t)g.batch_idx[(int64_t)blockIdx.x * kB + tid];
        else return (int64_t)g.batch_idx[(int64_t)(it_ < g.n_iters ? it_ : g.n_iters - 1) * kB + tid];
    };
    if (tid < kB) {
        fetch_rows(rec0);
        if constexpr (!TILED) rec_next = idx_of(1);
    }
    __syncthreads();   // parameter images complete

    float r_cur = 0.0f, t_cur = 0.0f;
    for (int it = 0; it < g.n_iters; ++it) {
#ifdef SSC_DDPG_DIAG
        uint64_t cp_prev = __builtin_amdgcn_s_memtime();
#endif
        // ---- L0: ReplayBuffer.sample_batch rows; MpiAdam step sizes --------------------------------------------------
        ++tA; ++tC;
        b1a *= (double)d.beta1; b2a *= (double)d.beta2; b1c *= (double)d.beta1; b2c *= (double)d.beta2;
        if (!TILED && tid == kB) {
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
            if constexpr (!TILED) {
                fetch_rows(rec_next);        // rows of iteration it + 1: in flight until the next L0
                rec_next = idx_of(it + 2);
            }
        }
        lds_barrier();
        LEVEL_MARK(0);

        // ---- L1: layer 1 of the four nets, ONE k-step of the fp32 MFMA per 16 x 16 tile: k = the O inputs, then the bias
        //      against a constant 1 (b1 follows W1 in the image, so A[k][unit] = img[64 k + unit] for k <= O), zeros above.
        //      Lane = sample with broadcast float4 weight reads was LDS-bound: 24 ds_read_b128 per lane -- a broadcast still
        //      takes its four LDS cycles -- against 8 ds_read_b32 now.  wave -> units 16 (wave >> 1).., samples 32 (wave & 1)..
        {
            static_assert(O + 1 <= 4, "inputs + bias fit one k-step");
            const int j1 = 16 * (wave >> 1), s1 = 32 * (wave & 1);
            const float one = (kg == O) ? 1.0f : 0.0f;
            float xb[2][2];     // [s2 / s][sample tile]: B[k][sample] = input k, 1 for k = O, 0 above
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float *X = row(q == 0 ? R::S2 : R::S) + (kg < O ? kg : 0) * kP + s1 + c;
                xb[q][0] = kg < O ? X[0] : one;
                xb[q][1] = kg < O ? X[16] : one;
            }
            const float *const img[4] = {th_ta, th_tc, th_c, th_a};   // W1 / b1 sit at the same offsets in both kinds of net
            float *const Z[4] = {row(R::T1), row(R::X2B), row(R::X2), row(R::U1)};
            float wa[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float v = img[n][NA::W1 + (kg <= O ? kg : 0) * H1 + j1 + c];
                wa[n] = kg <= O ? v : 0.0f;
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4m zero = {0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4m d0 = mfma4(wa[n], xb[n < 2 ? 0 : 1][0], zero), d1 = mfma4(wa[n], xb[n < 2 ? 0 : 1][1], zero);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *z = Z[n] + (j1 + 4 * kg + r) * kP + s1 + c;
                    z[0] = fmaxf(d0[r], 0.0f);
                    z[16] = fmaxf(d1[r], 0.0f);
                }
            }
        }
        lds_barrier();
        LEVEL_MARK(1);

        // ---- L2: layer 2 of target actor, actor, critic(s, a) -- and the 64 head rows of target critic layer 2, whose
        //      65th input (the target action) is the only part that has to wait for L3.  critic(s, pi(s)) in L4 shares
        //      its head with critic(s, a): same weights, same relu(layer 1) rows.
        //      wave -> (units 16 jb.., samples 32 sp..) of TWO nets: waves 0-3 target actor + the shared critic head,
        //      waves 4-7 actor + target critic head; the critic accumulators stay in registers until L4.
        const int jb = (wave >> 1) & 1, sp = wave & 1;
        const bool lo = wave < 4;
        const float *const img1 = lo ? th_ta : th_a, *const img2 = lo ? th_c : th_tc;
        TilePair keep;
        {
            const TilePair first = fwd_item(img1 + NA::W2, img1 + NA::b2, lo ? row(R::T1) : row(R::U1), jb, sp, c, kg);
            keep = fwd_item(img2 + NC::W2, img2 + NC::b2, lo ? row(R::X2) : row(R::X2B), jb, sp, c, kg);
            store_item<TANH2>(lo ? row(R::T2) : row(R::U2), first, nullptr, nullptr, jb, sp, c, kg);
            if (lo) store_item<TANH2>(row(R::CA2), keep, th_c + NC::W2 + H1 * W2S, row(R::X2 + H1), jb, sp, c, kg);   // + the action row
        }
        lds_barrier();
        LEVEL_MARK(2);

        // ---- L3: the three output layers, one wave each ------------------------------------------------------------
        if (wave == 0) row(R::X2B + H1)[b] = tanh_fast(dot32(th_ta + NA::W3, th_ta + NA::b3, row(R::T2), b));   // pi'(s2)  (:132)
        else if (wave == 1) row(R::PI)[b] = tanh_fast(dot32(th_a + NA::W3, th_a + NA::b3, row(R::U2), b));       // pi(s)    (:127)
        else if (wave == 2) row(R::Q)[b] = dot32(th_c + NC::W3, th_c + NC::b3, row(R::CA2), b);                  // Q(s, a)  (:181)
        lds_barrier();
        LEVEL_MARK(3);

        // ---- L4: layer 2 of target critic(s2, pi'(s2)) and critic(s, pi(s)): the action rows onto the heads of L2 ----
        store_item<TANH2>(lo ? row(R::CB2) : row(R::TB2), keep, img2 + NC::W2 + H1 * W2S, lo ? row(R::PI) : row(R::X2B + H1), jb, sp, c, kg);
        lds_barrier();
        LEVEL_MARK(4);

        // ---- L5 -------------------------------------------------------------------------------------------------------
        if (wave == 0) {          // target_Q = r + (1 - terminal) gamma Q'(s2, pi'(s2))  (:132-133); critic loss = mean((Q - y)^2)  (:181)
            const float qt = dot32(th_tc + NC::W3, th_tc + NC::b3, row(R::TB2), b);
            const float y = r_cur + (1.0f - t_cur) * d.gamma * qt;
            const float e = row(R::Q)[b] - y;
            row(R::DQ)[b] = 2.0f * e * inv_b;
            row(R::LC)[b] = e * e;
        } else if (wave == 1) {   // actor loss = -mean Q(s, pi(s))  (:168)
            row(R::LA)[b] = -dot32(th_c + NC::W3, th_c + NC::b3, row(R::CB2), b);
        } else {                  // (s, pi) path: dq = -1/B, delta of critic layer 2 = W3 dq act2'(.)
            for (int e = tid - 2 * 64; e < H2 * kB; e += kT - 2 * 64) {
                const int o = (e >> 6) * kP + (e & 63);
                row(R::DZB2)[o] = th_c[NC::W3 + (e >> 6)] * (-inv_b) * act2_deriv<TANH2>(row(R::CB2)[o]);
            }
        }
        lds_barrier();
        LEVEL_MARK(5);

        // ---- L6: TD path delta of critic layer 2; d(-Q)/d(action) through the output tanh ---------------------------
#pragma unroll
        for (int q = 0; q < H2 * kB / kT; ++q) {
            const int e = tid + kT * q, o = (e >> 6) * kP + (e & 63);
            row(R::DZ2)[o] = th_c[NC::W3 + (e >> 6)] * row(R::DQ)[e & 63] * act2_deriv<TANH2>(row(R::CA2)[o]);
        }
        if (wave == 1 || wave == 2) {   // the loss sums: six dependent cross-lane steps each, on waves with nothing else to do here
            const float sum = wave_sum(row(wave == 1 ? R::LC : R::LA)[b]);
            if (b == 0) red[wave - 1] = sum;
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
        LEVEL_MARK(6);

        // ---- L7: critic layer-1 delta (MFMA: wave -> 16 input rows x 32 samples); actor layer-2 delta ------------------
        bwd_item(th_c + NC::W2, row(R::DZ2), row(R::X2), row(R::DZ1), wave >> 1, wave & 1, c, kg);
#pragma unroll
        for (int q = 0; q < H2 * kB / kT; ++q) {
            const int e = tid + kT * q, o = (e >> 6) * kP + (e & 63);
            row(R::DZ2A)[o] = th_a[NA::W3 + (e >> 6)] * row(R::DZ3A)[e & 63] * act2_deriv<TANH2>(row(R::U2)[o]);
        }
        lds_barrier();
        LEVEL_MARK(7);

        // ---- L8: actor layer-1 delta; critic gradients -> MpiAdam -> target critic -----------------------------------
        bwd_item(th_a + NA::W2, row(R::DZ2A), row(R::U1), row(R::DZ1A), wave >> 1, wave & 1, c, kg);
        LEVEL_MARK(10);
        {
            const f32x4m gw = wgrad_tile(row(R::X2), row(R::DZ2), w_ib, w_jb, c, kg);
            LEVEL_MARK(11);
            if constexpr (TILED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gout[NA::gsize + NC::gW2 + (16 * w_ib + 4 * kg + r) * H2 + 16 * w_jb + c] = gw[r];
                if (ec.on) gout[NA::gsize + ec.gidx] = small_grad(lds, ec);
            } else {
                const AdamCfg cc = cfg_s[1];
                adam_tile(th_c, th_tc, NC::W2 + (16 * w_ib + 4 * kg) * W2S + 16 * w_jb + c, gw, mW2c, vW2c, cc, d.tau);
                LEVEL_MARK(12);
                if (ec.on) {
                    float th = th_c[ec.lidx], tg = th_tc[ec.lidx];
                    adam_target(th, tg, mc, vc, small_grad(lds, ec), cc, d.tau);
                    th_c[ec.lidx] = th; th_tc[ec.lidx] = tg;
                }
            }
        }
        LEVEL_MARK(13);
        lds_barrier();
        LEVEL_MARK(8);

        // ---- L9: actor gradients -> MpiAdam -> target actor; losses ------------------------------------------------------
        {
            const f32x4m gw = wgrad_tile(row(R::U1), row(R::DZ2A), w_ib, w_jb, c, kg);
            if constexpr (TILED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gout[NA::gW2 + (16 * w_ib + 4 * kg + r) * H2 + 16 * w_jb + c] = gw[r];
                if (ea.on) gout[ea.gidx] = small_grad(lds, ea);
            } else {
                const AdamCfg ca = cfg_s[0];
                adam_tile(th_a, th_ta, NA::W2 + (16 * w_ib + 4 * kg) * W2S + 16 * w_jb + c, gw, mW2a, vW2a, ca, d.tau);
                if (ea.on) {
                    float th = th_a[ea.lidx], tg = th_ta[ea.lidx];
                    adam_target(th, tg, ma, va, small_grad(lds, ea), ca, d.tau);
                    th_a[ea.lidx] = th; th_ta[ea.lidx] = tg;
                }
            }
        }
#ifndef SSC_DDPG_DIAG
        if constexpr (TILED) {
            if (tid == 0) { g.lpart[2 * blockIdx.x + 0] = red[0]; g.lpart[2 * blockIdx.x + 1] = red[1]; }
        } else if (tid == 0 && g.losses != nullptr) {
            g.losses[2 * it + 0] = red[0] / (float)kB;
            g.losses[2 * it + 1] = red[1] / (float)kB;
        }
#endif
        lds_barrier();
        LEVEL_MARK(9);
    }

    if constexpr (TILED) return;
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

template <int O, bool TANH2, bool TILED = false>
int launch_fixed(const FixedArgs &g, hipStream_t stream, unsigned tiles = 1) {
    const size_t lds = ((size_t)Rows<O>::total * kP + 2 * (Net<O, H1>::size + Net<O, H1 + 1>::size)) * sizeof(float);
    static_assert(((size_t)Rows<O>::total * kP + 2 * (Net<O, H1>::size + Net<O, H1 + 1>::size)) * sizeof(float) <= 160 * 1024 - 256,
                  "activation rows + parameter images must fit the LDS");
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(ddpg_train_fixed_kernel<O, TANH2, TILED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "hipFuncSetAttribute(ddpg_train_fixed_kernel)");
    if (rc) return rc;
    hipLaunchKernelGGL((ddpg_train_fixed_kernel<O, TANH2, TILED>), dim3(tiles), dim3(kT), lds, stream, g);
    return check_launch("ssc_ddpg_train");
}

}  // namespace

bool ddpg_fixed_shape(const ssc_ddpg_desc *d) {
    return d->batch_size == kB && (d->obs_dim == 2 || d->obs_dim == 3) && d->act_dim == 1 && d->actor_h1 == H1 && d->actor_h2 == H2 &&
           d->critic_h1 == H1 && d->critic_h2 == H2;
}

static bool fixed_nets(const ssc_ddpg_desc *d) {
    return (d->obs_dim == 2 || d->obs_dim == 3) && d->act_dim == 1 && d->actor_h1 == H1 && d->actor_h2 == H2 && d->critic_h1 == H1 &&
           d->critic_h2 == H2 && !d->layer_norm;
}

bool ddpg_fixed_tiled_shape(const ssc_ddpg_desc *d) {
    return fixed_nets(d) && d->batch_size > kB && d->batch_size % kB == 0 && d->batch_size <= 4096;
}

// arguments already validated by ssc_ddpg_train_ws; the workspace is sized by ddpg_wide_workspace_bytes (16-row partials:
// four times what the 64-row tiles write)
int ddpg_train_fixed_tiled(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                           float *d_losses, void *d_workspace, size_t workspace_bytes, hipStream_t stream) {
    const size_t need = ddpg_wide_workspace_bytes(d);
    SSC_REQUIRE(d_workspace != nullptr && workspace_bytes >= need,
                "ssc_ddpg_train_ws: workspace %zu < %zu bytes (ssc_ddpg_train_workspace_bytes)", workspace_bytes, need);
    const int tiles = d->batch_size / kB;
    const WidePartials wp = ddpg_wide_partials(d, d_workspace, tiles);
    FixedArgs g{*d, *rp, d_batch_idx, nullptr, 1, 1.0f / (float)d->batch_size, wp.gpart, wp.lpart};
    const bool t2 = d->last_layer_tanh != 0;
    for (int it = 0; it < n_iters; ++it) {
        g.batch_idx = d_batch_idx + (int64_t)it * d->batch_size;
        int rc;
        if (d->obs_dim == 2) rc = t2 ? launch_fixed<2, true, true>(g, stream, tiles) : launch_fixed<2, false, true>(g, stream, tiles);
        else rc = t2 ? launch_fixed<3, true, true>(g, stream, tiles) : launch_fixed<3, false, true>(g, stream, tiles);
        if (rc) return rc;
        ddpg_wide_apply(d, d_workspace, tiles, it, d_losses ? d_losses + 2 * it : nullptr, stream);
    }
    ddpg_wide_finish(d, n_iters, stream);
    return check_launch("ssc_ddpg_train_ws");
}

// arguments already validated by ssc_ddpg_train
int ddpg_train_fixed(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                     float *d_losses, hipStream_t stream) {
    FixedArgs g{*d, *rp, d_batch_idx, d_losses, n_iters, 1.0f / (float)kB, nullptr, nullptr};
    const bool t2 = d->last_layer_tanh != 0;
    if (d->obs_dim == 2) return t2 ? launch_fixed<2, true>(g, stream) : launch_fixed<2, false>(g, stream);
    return t2 ? launch_fixed<3, true>(g, stream) : launch_fixed<3, false>(g, stream);
}

}  // namespace ssc
