// ddpg_train.hip -- the DDPG learner step on the GPU: DDPG_editted.train() + update_target_net()
// (DDPG_Baselines_editted/ddpg_editted.py:287-339; graph :127-133,168-199) for batch 64 and the
// 64-32 actor / critic of the shipped runs.
//
// The iterations form a serial chain through the parameters (the reference runs one per env step),
// so ONE workgroup executes `n_iters` of them per launch.  Design:
//   * everything lives in LDS for the whole launch: the four parameter vectors (actor, critic, their
//     targets; written back once at the end) and every activation / delta of the batch as [unit][68]
//     rows; only the Adam moments stay in global memory (one coalesced pass per iteration);
//   * an iteration is a LIST OF STEPS (built on the host, ~40 entries: dense forward, dense backward,
//     activation derivative, weight gradient + Adam, and a few special ones) run by one interpreter loop,
//     so each dense routine exists ONCE in the binary.  The first versions inlined the dense code at
//     every call site: 15 k -- later 90 k -- instructions, several times the 64 KB instruction cache, and the
//     kernel was bound by instruction fetch (70 us per iteration whatever the data path looked like);
//   * dense passes use LANE = SAMPLE (64 lanes busy whatever the layer width); the output units
//     (forward) or input rows (backward) are dealt to the 4 waves in groups of 4, and a weight is one
//     broadcast LDS read -- a float4 of 4 consecutive units where the row length allows.
// fp32, plain FMA chains in k order; MpiAdam's bias-corrected step size is computed in f64.
#include "ddpg_device.h"
#include "ssc_host.h"
#include <cstdlib>

namespace ssc {

#ifndef SSC_DDPG_THREADS
#define SSC_DDPG_THREADS 512
#endif
constexpr int kTrainThreads = SSC_DDPG_THREADS;
constexpr int kNW = kTrainThreads / 64;   // waves: the dense passes deal unit groups / input rows round-robin to them
constexpr int kMaxSteps = 44;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };
enum {
    ST_GATHER = 0,   // batch rows -> S, S2, action rows of X2, r, terminal
    ST_FWD,          // Z = act(b + W^T X)                       w, b, in, out, x, z, act, xtail, tail_from
    ST_YTARGET,      // y = r + (1 - t) gamma q'                 x = q' row
    ST_CLOSS,        // e = q - y; closs; dq = 2 e / B           x = q row, z = dq row
    ST_FILL,         // rows z = constant c                      z, out rows, c
    ST_BWD,          // dX[i - i0] = act'(A[i - i0]) sum_j W[i][j] dZ[j]    w, out, i0, i1, x = dZ, z = dX, act, xtail = A
    ST_DERIV,        // D[j] *= act'(A[j])                       x = A, z = D, out rows, act
    ST_ALOSS,        // aloss = -q(s, pi(s))                     x = q row
    ST_LOSSES,       // block reduction of the two losses, Adam step sizes
    ST_WGRAD,        // dW = X dZ^T, db; Adam into theta         w, b, in, out, x = X, z = dZ, net, xtail, tail_from
    ST_TUPDATE,      // theta' <- (1 - tau) theta' + tau theta, losses out
};

// One step.  x, z, xtail (and w, b of the dense steps) are offsets in floats from the start of the dynamic LDS
// array; for ST_WGRAD w, b index the flat parameter / moment arrays of the net.
struct alignas(16) Step {
    int16_t kind, barrier;          // barrier: __syncthreads() after the step
    int16_t in, out, act, net;      // net: 0 actor / 1 critic (ST_WGRAD)
    int16_t i0, i1, tail_from, pad;
    int32_t w, b, x, z, xtail;
    float c;
    int32_t m_lds;   // ST_WGRAD: LDS offset of this layer's Adam moments ([W | b] first moments, then the second ones), or -1: global
};

static_assert(sizeof(Step) == 48, "a step is three 16-byte scalar loads");

struct TrainArgs {   // passed by value: must stay below the 4 KB kernel-argument limit
    ssc_ddpg_desc d;
    ssc_replay_view rp;
    const int32_t *batch_idx;
    float *losses;
    int32_t n_iters, n_steps;
    int32_t off_theta[4];   // LDS offsets of actor, critic, target actor, target critic
    int32_t n_actor, n_critic;
    int32_t off_S, off_S2, off_RT, off_X2act;   // gather targets: state rows, next-state rows, (r, t, y) rows, action rows
    Step steps[kMaxSteps];
};
static_assert(sizeof(TrainArgs) <= 4000, "TrainArgs must fit the kernel-argument segment");

#ifdef SSC_DDPG_DIAG
// intra-step marks of thread 0 (cycles since the step began), copied out for step SSC_DDPG_DIAG_STEP
#ifndef SSC_DDPG_DIAG_STEP
#define SSC_DDPG_DIAG_STEP 10
#endif
__shared__ uint64_t diag_t0;
__shared__ float diag_marks[12];
#define DIAG_MARK(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) diag_marks[k] = (float)(__builtin_amdgcn_s_memtime() - diag_t0); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DIAG_MARK(k) do { } while (0)
#endif

template <bool AL4>
__device__ __forceinline__ f4 weights4(const float *w, int j, int out) {
    if (AL4) return *reinterpret_cast<const f4 *>(w + j);
    f4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (j + e < out) ? w[j + e] : 0.0f;
    return r;
}

// (tanh on the transcendental pipe, |err| ~ 1e-7: ocml's tanhf is ~150 instructions per inlined call site, and the whole
// kernel has to stay well inside the 64 KB instruction cache: it was 70 KB, it is 27 KB now)
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_TANH) return tanh_fast(v);
    return v;
}

// Z[j][b] = act(bias[j] + sum_i W[i][j] X[i][b]) for the NQ unit groups {4 (wave + kNW q)} of this wave.
// Input rows i >= tail_from come from Xtail (the critic's action rows live apart from relu(layer 1)).
template <int NQ, bool AL4>
__device__ __forceinline__ void dense_fwd_groups(const float *W, const float *bias, int in, int out, const float *X,
                                                 float *Z, int act, const float *Xtail, int tail_from) {
    const int wave = threadIdx.x >> 6, b = threadIdx.x & 63;
    f4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = weights4<AL4>(bias, 4 * (wave + kNW * q), out);
    const int n_head = in < tail_from ? in : tail_from;
#pragma unroll 2
    for (int i = 0; i < n_head; ++i) {
        const float x = X[i * kP + b];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] += weights4<AL4>(W + i * out, 4 * (wave + kNW * q), out) * x;
    }
    for (int i = n_head; i < in; ++i) {
        const float x = Xtail[(i - tail_from) * kP + b];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] += weights4<AL4>(W + i * out, 4 * (wave + kNW * q), out) * x;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * (wave + kNW * q) + e;
            if (AL4 || j < out) Z[j * kP + b] = apply_act(acc[q][e], act);
        }
}

// out == 1 (the critic's Q, a 1-d action): W is a column of `in` floats; wave 0 walks it 4 inputs per float4.
__device__ __forceinline__ void dense_fwd_out1(const float *W, const float *bias, int in, const float *X, float *Z, int act) {
    if (threadIdx.x >= 64) return;
    const int b = threadIdx.x;
    float acc = bias[0];
    int i = 0;
    if ((reinterpret_cast<uintptr_t>(W) & 15) == 0) {
#pragma unroll 4
        for (; i + 4 <= in; i += 4) {
            const f4 w = *reinterpret_cast<const f4 *>(W + i);
            acc += (w[0] * X[i * kP + b] + w[1] * X[(i + 1) * kP + b]) + (w[2] * X[(i + 2) * kP + b] + w[3] * X[(i + 3) * kP + b]);
        }
    }
    for (; i < in; ++i) acc += W[i] * X[i * kP + b];
    Z[b] = apply_act(acc, act);
}

template <bool AL4>
__device__ __forceinline__ void dense_fwd(const float *W, const float *bias, int in, int out, const float *X, float *Z,
                                          int act, const float *Xtail, int tail_from) {
    const int wave = threadIdx.x >> 6;
    const int groups = (out + 3) >> 2;
    const int nq = (groups - wave + kNW - 1) / kNW;   // groups wave, wave + kNW, ... < groups  (out <= 64: nq <= 2)
    // (the wide layers run on the MFMA: what is left here are layer 1 with its 2..8 inputs and ragged sizes, so ONE small
    // instance per alignment beats four unrolled ones -- code size)
    if (nq >= 2) dense_fwd_groups<2, AL4>(W, bias, in, out, X, Z, act, Xtail, tail_from);
    else if (nq == 1) dense_fwd_groups<1, AL4>(W, bias, in, out, X, Z, act, Xtail, tail_from);
}

// dX[i - i0][b] = sum_j W[i][j] dZ[j][b] for the NR input rows {i0 + wave + kNW (qbase + r)} of this wave.
// act != ACT_NONE: the result is multiplied by act'(A[i - i0]) (tanh: 1 - a^2, relu: a > 0) on the way out.
__device__ __forceinline__ float act_deriv(float a, int act) {
    return act == ACT_TANH ? 1.0f - a * a : (act == ACT_RELU ? (a > 0.0f ? 1.0f : 0.0f) : 1.0f);
}

template <int NR, bool AL4>
__device__ __forceinline__ void dense_bwd_rows(const float *W, int out, int i0, int qbase, const float *dZ, float *dX,
                                               const float *A, int act) {
    const int wave = threadIdx.x >> 6, b = threadIdx.x & 63;
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0f;
#pragma unroll 2
    for (int j = 0; j < out; j += 4) {
        f4 dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = (AL4 || j + e < out) ? dZ[(j + e) * kP + b] : 0.0f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = i0 + wave + kNW * (qbase + r);
            const f4 w = weights4<AL4>(W + i * out, j, out);
            acc[r] += (w[0] * dz[0] + w[1] * dz[1]) + (w[2] * dz[2] + w[3] * dz[3]);
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int o = (wave + kNW * (qbase + r)) * kP + b;
        dX[o] = acc[r] * act_deriv(A[o], act);
    }
}

// out == 1 (the layer above is the critic's Q or a 1-d action): dX[i - i0][b] = W[i] dZ[0][b] act'(A[i - i0][b]),
// elementwise over (i1 - i0) x 64 values -- the generic routine spent 2.7 k cycles on its per-element bounds checks.
__device__ __forceinline__ void dense_bwd_out1(const float *W, int i0, int i1, const float *dZ, float *dX, const float *A, int act) {
    for (int e = threadIdx.x; e < (i1 - i0) * kB; e += kTrainThreads) {
        const int r = e >> 6, b = e & 63, o = r * kP + b;
        dX[o] = W[i0 + r] * dZ[b] * act_deriv(A[o], act);
    }
}

template <bool AL4>
__device__ __forceinline__ void dense_bwd_in(const float *W, int out, int i0, int i1, const float *dZ, float *dX,
                                             const float *A, int act) {
    const int wave = threadIdx.x >> 6;
    const int nr = (i1 - i0 - wave + kNW - 1) / kNW;   // rows i0 + wave, i0 + wave + kNW, ... of this wave
    int q = 0;
    for (; q + 2 <= nr; q += 2) dense_bwd_rows<2, AL4>(W, out, i0, q, dZ, dX, A, act);
    if (nr - q == 1) dense_bwd_rows<1, AL4>(W, out, i0, q, dZ, dX, A, act);
}

// ---- the wide contractions on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) ------------------------------------------
// Lane = sample with one broadcast weight per FMA makes EVERY wave re-read the whole activation block from LDS for its
// 4 output units: a 64 -> 32 layer moved 139 KB through the LDS and took ~6 k cycles, the pass through W2 11 k, its
// gradient 13 k -- together 60 % of an iteration.  As 16 x 16 tiles of D = A B (A [16 x 4], B [4 x 16] per k-step, one
// register per lane each, D four) the same contractions read every operand element once per tile: one or two tiles per
// wave.  Products and sums are exact fp32 (the instruction is an fp32 FMA chain), so the fp64-oracle tolerance does not
// move.  The 32x32x2 shape was measured in round 1 (no gain: a 64 -> 32 layer is only TWO 32 x 32 tiles, six of eight
// waves idle); 16 x 16 tiles give every wave one.  Only layers whose sizes are multiples of 16 come here: a version
// with operand guards for every size (layer 1, the 1-unit outputs, ragged nets) was built and measured SLOWER on those
// (41.8 vs 32.8 us per iteration) -- the small layers stay on the VALU routines above.
// Lane l of a wave: A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15], D[row 4 (l >> 4) + r][col l & 15].

// acc += sum over `nsteps` k-steps of A-fragment x B-fragment (operand element of step s at ap[s * astride], bp[s * bstride]).
// The operands of U steps are requested together and the U MFMAs issued behind them: hipcc does not unroll the plain
// loop (runtime trip count), which left every MFMA waiting on its own LDS round trip -- 16 serial round trips for a
// 64-deep contraction, ~2 k of the ~3.3 k cycles such a step took.
template <int U>
__device__ __forceinline__ void mfma_chain(f32x4m &acc, const float *ap, int astride, const float *bp, int bstride, int nsteps) {
    int s = 0;
    for (; s + U <= nsteps; s += U) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = ap[(s + u) * astride]; b[u] = bp[(s + u) * bstride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma4(a[u], b[u], acc);
    }
    for (; s < nsteps; ++s) acc = mfma4(ap[s * astride], bp[s * bstride], acc);
}

// forward: Z[j][b] = act(bias[j] + sum_i W[i][j] X[i][b]);  rows = units j, cols = samples b, k = inputs i.
// Needs out % 16 == 0; input rows beyond the last multiple of 4 (the critic's 65th row: the action) and the rows held
// in Xtail are added on the VALU to the accumulators.
__device__ __forceinline__ void dense_fwd_mfma(const float *W, const float *bias, int in, int out, const float *X, float *Z,
                                               int act, const float *Xtail, int tail_from) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int n_head = in < tail_from ? in : tail_from;
    const int n_tiles = (out >> 4) * (kB / 16);
    for (int tile = wave; tile < n_tiles; tile += kNW) {
        const int j0 = (tile >> 2) * 16, b0 = (tile & 3) * 16;
        DIAG_MARK(1);
        f32x4m acc = *reinterpret_cast<const f32x4m *>(bias + j0 + 4 * kg);
        const float *wp = W + kg * out + j0 + c, *xp = X + kg * kP + b0 + c;
        const int n4 = n_head & ~3;
        mfma_chain<8>(acc, wp, 4 * out, xp, 4 * kP, n4 >> 2);
        DIAG_MARK(2);
        for (int i = n4; i < in; ++i) {
            const float x = (i < tail_from) ? X[i * kP + b0 + c] : Xtail[(i - tail_from) * kP + b0 + c];
            acc += *reinterpret_cast<const f32x4m *>(W + i * out + j0 + 4 * kg) * x;
        }
        DIAG_MARK(3);
        if (act == ACT_RELU) {          // (one wave-uniform branch per tile, not one per element)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.0f);
        } else if (act == ACT_TANH) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = tanh_fast(acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[(j0 + 4 * kg + r) * kP + b0 + c] = acc[r];
        DIAG_MARK(4);
    }
}

// backward: dX[i - i0][b] = act'(A[i - i0][b]) sum_j W[i][j] dZ[j][b];  rows = input rows i, cols = samples, k = units j.
// Needs (i1 - i0) % 16 == 0 and out % 4 == 0.  A wave takes one 16-row block and TWO sample tiles, so the A fragments
// (a column slice of row-major W: a strided, bank-conflicted read) are fetched once per pair.
__device__ __forceinline__ void dense_bwd_mfma(const float *W, int out, int i0, int i1, const float *dZ, float *dX,
                                               const float *A, int act) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int n_items = ((i1 - i0) >> 4) * 2;     // (row block, pair of sample tiles)
    for (int item = wave; item < n_items; item += kNW) {
        const int ib = item >> 1, sb = (item & 1) * 2;
        f32x4m acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
        const float *wp = W + (i0 + ib * 16 + c) * out + kg;
        const float *z0 = dZ + kg * kP + sb * 16 + c, *z1 = z0 + 16;
        int s4 = 0;
        const int nsteps = out >> 2;
        for (; s4 + 4 <= nsteps; s4 += 4) {   // 4 k-steps: 12 operand reads in flight, then 8 MFMAs on two accumulators
            float a[4], u0[4], u1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = wp[4 * (s4 + u)]; u0[u] = z0[4 * (s4 + u) * kP]; u1[u] = z1[4 * (s4 + u) * kP]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc0 = mfma4(a[u], u0[u], acc0); acc1 = mfma4(a[u], u1[u], acc1); }
        }
        for (; s4 < nsteps; ++s4) {
            const float a = wp[4 * s4];
            acc0 = mfma4(a, z0[4 * s4 * kP], acc0);
            acc1 = mfma4(a, z1[4 * s4 * kP], acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (ib * 16 + 4 * kg + r) * kP + sb * 16 + c;
            dX[o] = acc0[r] * act_deriv(A[o], act);
            dX[o + 16] = acc1[r] * act_deriv(A[o + 16], act);
        }
    }
}


// MpiAdam.update (baselines common/mpi_adam.py [third-party], ddpg_editted.py:326-327): theta in LDS, moments in
// global memory, same index.
__device__ __forceinline__ void adam_apply(float *theta, float *m, float *v, int idx, float g, const AdamCfg &c) {
    const float mi = c.beta1 * m[idx] + (1.0f - c.beta1) * g;
    const float vi = c.beta2 * v[idx] + (1.0f - c.beta2) * (g * g);
    m[idx] = mi;
    v[idx] = vi;
    theta[idx] += (-c.a) * mi * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vi) + c.eps);   // 1-ulp sqrt / rcp: a 1e-7 relative
}                                                                                            // wobble of a 1e-3-sized step

// dW[i][j] = sum_b X[i][b] dZ[j][b]; db[j] = sum_b dZ[j][b]; applied straight into Adam.
// (Fetching the moments of 4 elements ahead of their dot products was measured 2x SLOWER: the pass is bound by
// the LDS reads of the dot products, not by the global latency of the moments.)
__device__ __forceinline__ void weight_grad_adam(const float *X, const float *dZ, int in, int out, float *theta,
                                                 float *m, float *v, int offW, int offb, const AdamCfg &c,
                                                 const float *Xtail, int tail_from) {
    int first = 0;   // rows [0, first) of dW are done on the MFMA below
    const int n_head = in < tail_from ? in : tail_from;
    if ((out & 15) == 0 && n_head >= 16) {
        // dW[i][j] = sum_b X[i][b] dZ[j][b]: rows = input rows i, cols = units j, k = samples b (64 = 16 k-steps); the
        // accumulator goes straight into Adam (4 elements per lane, 16 consecutive j per k group: coalesced moments)
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cc = lane & 15, kg = lane >> 4;
        first = n_head & ~15;
        const int jt = out >> 4, n_tiles = (first >> 4) * jt;
        for (int tile = wave; tile < n_tiles; tile += kNW) {
            const int ib = tile / jt, jb = tile - ib * jt;
            const float *xp = X + (ib * 16 + cc) * kP + kg, *zp = dZ + (jb * 16 + cc) * kP + kg;
            f32x4m acc = {0.0f, 0.0f, 0.0f, 0.0f};
            float m0[4], v0[4];   // the moments are requested BEFORE the contraction: their global round trip runs under it
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = offW + (ib * 16 + 4 * kg + r) * out + jb * 16 + cc;
                m0[r] = m[idx];
                v0[r] = v[idx];
            }
            mfma_chain<8>(acc, xp, 4, zp, 4, kB / 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = offW + (ib * 16 + 4 * kg + r) * out + jb * 16 + cc;
                const float mi = c.beta1 * m0[r] + (1.0f - c.beta1) * acc[r];
                const float vi = c.beta2 * v0[r] + (1.0f - c.beta2) * (acc[r] * acc[r]);
                m[idx] = mi;
                v[idx] = vi;
                theta[idx] += (-c.a) * mi * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vi) + c.eps);
            }
        }
    }
    for (int idx = first * out + threadIdx.x; idx < in * out; idx += kTrainThreads) {
        const int i = idx / out, j = idx - i * out;
        const float *xr = (i >= tail_from) ? Xtail + (i - tail_from) * kP : X + i * kP;
        f4 g4 = (f4)(0.0f);
#pragma unroll 4
        for (int q = 0; q < kB / 4; ++q)
            g4 += *reinterpret_cast<const f4 *>(xr + 4 * q) * *reinterpret_cast<const f4 *>(dZ + j * kP + 4 * q);
        adam_apply(theta, m, v, offW + idx, (g4[0] + g4[1]) + (g4[2] + g4[3]), c);
    }
    for (int j = threadIdx.x; j < out; j += kTrainThreads) {
        f4 g4 = (f4)(0.0f);
        for (int q = 0; q < kB / 4; ++q) g4 += *reinterpret_cast<const f4 *>(dZ + j * kP + 4 * q);
        adam_apply(theta, m, v, offb + j, (g4[0] + g4[1]) + (g4[2] + g4[3]), c);
    }
}

// Diagnostic build (-DSSC_DDPG_DIAG, tools/exp_ddpg_phases.py): d_losses is [n_iters][kMaxSteps] and receives
// the cycles thread 0 spent in each step instead of the losses.
__global__ __launch_bounds__(kTrainThreads) void ddpg_train_kernel(TrainArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[2][kNW];
    __shared__ AdamCfg cfg_s[2];
    const ssc_ddpg_desc &d = g.d;
    const int tid = threadIdx.x, b = tid & 63;
    float *th_a = lds + g.off_theta[0], *th_c = lds + g.off_theta[1];
    float *th_ta = lds + g.off_theta[2], *th_tc = lds + g.off_theta[3];
    for (int e = tid; e < g.n_actor; e += kTrainThreads) { th_a[e] = d.actor[e]; th_ta[e] = d.target_actor[e]; }
    for (int e = tid; e < g.n_critic; e += kTrainThreads) { th_c[e] = d.critic[e]; th_tc[e] = d.target_critic[e]; }
    // Adam moments of the SMALL layers live in LDS for the whole launch (their update steps are a few hundred elements
    // each: with the moments in global memory every one of them was two dependent memory round trips)
    for (int si = 0; si < g.n_steps; ++si) {
        const Step &st = g.steps[si];
        if (st.kind != ST_WGRAD || st.m_lds < 0) continue;
        const int cnt = st.in * st.out + st.out;
        const float *gm = st.net == 0 ? d.adam_m_actor : d.adam_m_critic, *gv = st.net == 0 ? d.adam_v_actor : d.adam_v_critic;
        for (int e = tid; e < cnt; e += kTrainThreads) {
            lds[st.m_lds + e] = gm[st.w + e];
            lds[st.m_lds + cnt + e] = gv[st.w + e];
        }
    }
    __syncthreads();
    int tA = d.adam_t[0], tC = d.adam_t[1];
    float closs = 0.0f;
    // running beta powers for MpiAdam's bias correction, in f64 (1 - 0.999^t loses 5 digits in fp32): one ipow per
    // launch, then one multiplication per iteration -- the per-iteration ipow + sqrt + divisions sat on the critical
    // path (3.6 k cycles with every other thread waiting at the barrier)
    double b1a = ipow((double)d.beta1, tA), b2a = ipow((double)d.beta2, tA);
    double b1c = ipow((double)d.beta1, tC), b2c = ipow((double)d.beta2, tC);
    // (requesting the batch rows of iteration it + 1 in the middle of iteration it and parking them in registers was
    // tried: hipcc drains every outstanding load in front of the next __syncthreads(), so the round trips only moved)
    float pf_s[SSC_MAX_STATE], pf_s2[SSC_MAX_STATE], pf_a[SSC_MAX_ACT], pf_r = 0.0f, pf_t = 0.0f;
    auto fetch_rows = [&](int it_) {
        const int64_t rec = g.batch_idx[(int64_t)it_ * kB + tid];
#pragma unroll
        for (int c = 0; c < SSC_MAX_STATE; ++c)
            if (c < d.obs_dim) { pf_s[c] = g.rp.s[rec * d.obs_dim + c]; pf_s2[c] = g.rp.s2[rec * d.obs_dim + c]; }
#pragma unroll
        for (int c = 0; c < SSC_MAX_ACT; ++c)
            if (c < d.act_dim) pf_a[c] = g.rp.a[rec * d.act_dim + c];
        pf_r = g.rp.r[rec];
        pf_t = g.rp.t[rec] ? 1.0f : 0.0f;
    };

    // A step descriptor is fetched WHOLE (three s_load_dwordx4 from the kernel-argument segment) one step ahead of its
    // use.  Read field by field where needed, every step paid 4-5 DEPENDENT scalar-cache round trips (kind -> branch ->
    // sizes -> branch -> offsets ...), ~800 of the ~2000 cycles even the smallest step took.
    auto load_step = [&](int si_) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 *q = reinterpret_cast<const i32x4 *>(&g.steps[si_]);
        union { i32x4 v[3]; Step s; } u;
        u.v[0] = q[0]; u.v[1] = q[1]; u.v[2] = q[2];
        return u.s;
    };
    Step nxt = load_step(0);
    for (int it = 0; it < g.n_iters; ++it) {
#ifdef SSC_DDPG_DIAG
        uint64_t cp_prev = __builtin_amdgcn_s_memtime();
#endif
        for (int si = 0; si < g.n_steps; ++si) {
            const Step st = nxt;
            nxt = load_step(si + 1 < g.n_steps ? si + 1 : 0);
#ifdef SSC_DDPG_DIAG
            if (tid == 0) diag_t0 = cp_prev;
            DIAG_MARK(0);
#endif
            const float *X = lds + st.x;
            float *Z = lds + st.z;
            switch (st.kind) {
            case ST_GATHER:   // ReplayBuffer.sample_batch rows
                ++tA; ++tC;
                b1a *= (double)d.beta1; b2a *= (double)d.beta2; b1c *= (double)d.beta1; b2c *= (double)d.beta2;
                if (tid == kB) {   // a lane that has nothing to gather: the MpiAdam step sizes of this iteration
                    cfg_s[0] = AdamCfg{(float)((double)d.actor_lr * sqrt(1.0 - b2a) / (1.0 - b1a)), d.beta1, d.beta2, d.epsilon};
                    cfg_s[1] = AdamCfg{(float)((double)d.critic_lr * sqrt(1.0 - b2c) / (1.0 - b1c)), d.beta1, d.beta2, d.epsilon};
                }
                if (tid < kB) {
                    fetch_rows(it);
                    float *S = lds + g.off_S, *S2 = lds + g.off_S2, *RT = lds + g.off_RT, *XA = lds + g.off_X2act;
#pragma unroll
                    for (int c = 0; c < SSC_MAX_STATE; ++c)
                        if (c < d.obs_dim) {   // obs0 / obs1 enter every network clipped (ddpg_editted.py:106-109)
                            S[c * kP + tid] = d.obs_clip > 0.0f ? fminf(fmaxf(pf_s[c], -d.obs_clip), d.obs_clip) : pf_s[c];
                            S2[c * kP + tid] = d.obs_clip > 0.0f ? fminf(fmaxf(pf_s2[c], -d.obs_clip), d.obs_clip) : pf_s2[c];
                        }
#pragma unroll
                    for (int c = 0; c < SSC_MAX_ACT; ++c)
                        if (c < d.act_dim) XA[c * kP + tid] = pf_a[c];
                    RT[0 * kP + tid] = pf_r;
                    RT[1 * kP + tid] = pf_t;
                }
                break;
            case ST_FWD:
                if (st.out == 1 && st.tail_from >= st.in) {
                    dense_fwd_out1(lds + st.w, lds + st.b, st.in, X, Z, st.act);
                    // what used to be three interpreter steps of their own (each ~600 cycles of step overhead and, for two
                    // of them, a block barrier): the lane that just wrote q[b] finishes the per-sample arithmetic on it
                    if (st.i0 != 0 && tid < kB) {
                        const float q = Z[tid];
                        float *RT = lds + g.off_RT;
                        if (st.i0 == 1) {          // target_Q = r + (1 - terminal) * gamma * Q'(s2, pi'(s2))   (ddpg_editted.py:132-133)
                            RT[2 * kP + tid] = RT[tid] + (1.0f - RT[kP + tid]) * d.gamma * q;
                        } else if (st.i0 == 2) {   // critic loss = mean((Q - y)^2)  (:181); the row becomes d loss / d q
                            const float e = q - RT[2 * kP + tid];
                            closs = e * e;
                            Z[tid] = 2.0f * e / (float)kB;
                        } else {                   // actor loss = -mean Q(s, pi(s))  (:168); the row becomes its dq = -1/B
                            RT[3 * kP + tid] = -q;   // parked in LDS until ST_LOSSES
                            Z[tid] = -1.0f / (float)kB;
                        }
                    }
                }
                else if ((st.out & 15) == 0 && (st.in < st.tail_from ? st.in : st.tail_from) >= 16)
                    dense_fwd_mfma(lds + st.w, lds + st.b, st.in, st.out, X, Z, st.act, lds + st.xtail, st.tail_from);
                else if ((st.out & 3) == 0)
                    dense_fwd<true>(lds + st.w, lds + st.b, st.in, st.out, X, Z, st.act, lds + st.xtail, st.tail_from);
                else
                    dense_fwd<false>(lds + st.w, lds + st.b, st.in, st.out, X, Z, st.act, lds + st.xtail, st.tail_from);
                break;
            case ST_YTARGET:  // target_Q = r + (1 - terminal) * gamma * Q'(s2, pi'(s2))      (ddpg_editted.py:132-133)
                if (tid < kB) {
                    float *RT = lds + g.off_RT;
                    RT[2 * kP + tid] = RT[tid] + (1.0f - RT[kP + tid]) * d.gamma * X[tid];
                }
                break;
            case ST_CLOSS:    // critic loss = mean((Q - y)^2)                                  (:181)
                if (tid < kB) {
                    const float e = X[tid] - (lds + g.off_RT)[2 * kP + tid];
                    closs = e * e;
                    Z[tid] = 2.0f * e / (float)kB;   // d loss / d q
                }
                break;
            case ST_FILL:
                for (int e = tid; e < st.out * kB; e += kTrainThreads) Z[(e >> 6) * kP + (e & 63)] = st.c;
                break;
            case ST_BWD:
                if (st.out == 1)
                    dense_bwd_out1(lds + st.w, st.i0, st.i1, X, Z, lds + st.xtail, st.act);
                else if (((st.i1 - st.i0) & 15) == 0 && st.i1 > st.i0 && (st.out & 3) == 0 && st.out >= 16)
                    dense_bwd_mfma(lds + st.w, st.out, st.i0, st.i1, X, Z, lds + st.xtail, st.act);
                else if ((st.out & 3) == 0) dense_bwd_in<true>(lds + st.w, st.out, st.i0, st.i1, X, Z, lds + st.xtail, st.act);
                else dense_bwd_in<false>(lds + st.w, st.out, st.i0, st.i1, X, Z, lds + st.xtail, st.act);
                break;
            case ST_DERIV:    // delta *= act'(activation): tanh -> 1 - a^2, relu -> a > 0
                for (int e = tid; e < st.out * kB; e += kTrainThreads) {
                    const int o = (e >> 6) * kP + (e & 63);
                    const float a = X[o];
                    Z[o] *= (st.act == ACT_TANH) ? (1.0f - a * a) : (a > 0.0f ? 1.0f : 0.0f);
                }
                break;
            case ST_ALOSS:    // actor loss = -mean Q(s, pi(s))                                  (:168)
                if (tid < kB) (lds + g.off_RT)[3 * kP + tid] = -X[tid];
                break;
            case ST_LOSSES: {
                float v0 = (tid < kB) ? closs : 0.0f, v1 = (tid < kB) ? (lds + g.off_RT)[3 * kP + tid] : 0.0f;
#pragma unroll
                for (int msk = 32; msk >= 1; msk >>= 1) { v0 += __shfl_xor(v0, msk); v1 += __shfl_xor(v1, msk); }
                if (b == 0) { red[0][tid >> 6] = v0; red[1][tid >> 6] = v1; }
                break;
            }
            case ST_WGRAD: {  // gradients + MpiAdam, all from the OLD parameters' deltas         (:326-327)
                // two instances, so that each knows its address space: with one generic pointer the moment accesses were
                // flat_load / flat_store, which count on lgkmcnt as well -- every wait for an LDS operand of the contraction
                // then also waited for the global round trip of the prefetched moments
                float *theta = st.net == 0 ? th_a : th_c;
                const AdamCfg &cfg = cfg_s[st.net == 0 ? 0 : 1];
                if (st.m_lds >= 0) {   // flat index st.w + k of the layer maps to LDS word m_lds + k
                    float *mm = lds + (st.m_lds - st.w);
                    weight_grad_adam(X, Z, st.in, st.out, theta, mm, mm + (st.in * st.out + st.out), st.w, st.b, cfg,
                                     lds + st.xtail, st.tail_from);
                } else {
                    weight_grad_adam(X, Z, st.in, st.out, theta, st.net == 0 ? d.adam_m_actor : d.adam_m_critic,
                                     st.net == 0 ? d.adam_v_actor : d.adam_v_critic, st.w, st.b, cfg, lds + st.xtail, st.tail_from);
                }
                break;
            }
            case ST_TUPDATE:  // update_target_net: theta' <- (1 - tau) theta' + tau theta       (:338-339)
                for (int e = tid; e < g.n_actor; e += kTrainThreads) th_ta[e] = (1.0f - d.tau) * th_ta[e] + d.tau * th_a[e];
                for (int e = tid; e < g.n_critic; e += kTrainThreads) th_tc[e] = (1.0f - d.tau) * th_tc[e] + d.tau * th_c[e];
#ifndef SSC_DDPG_DIAG
                if (tid == 0 && g.losses != nullptr) {
                    float l0 = 0.0f, l1 = 0.0f;
                    for (int w = 0; w < kNW; ++w) { l0 += red[0][w]; l1 += red[1][w]; }
                    g.losses[2 * it + 0] = l0 / (float)kB;
                    g.losses[2 * it + 1] = l1 / (float)kB;
                }
#endif
                break;
            }
            DIAG_MARK(5);
            if (st.barrier) __syncthreads();
#ifdef SSC_DDPG_DIAG
            DIAG_MARK(6);
            if (si == SSC_DDPG_DIAG_STEP && tid == 0)
                for (int k = 0; k < 8; ++k) g.losses[(int64_t)kMaxSteps * it + 32 + k] = diag_marks[k];
            {
                __builtin_amdgcn_sched_barrier(0);
                const uint64_t now = __builtin_amdgcn_s_memtime();
                if (tid == 0) g.losses[(int64_t)kMaxSteps * it + si] = (float)(now - cp_prev);
                cp_prev = now;
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
    }
    // the forward kernels outside (ssc_actor_forward, ssc_critic_forward, rollouts) read the global arrays
    for (int e = tid; e < g.n_actor; e += kTrainThreads) { d.actor[e] = th_a[e]; d.target_actor[e] = th_ta[e]; }
    for (int e = tid; e < g.n_critic; e += kTrainThreads) { d.critic[e] = th_c[e]; d.target_critic[e] = th_tc[e]; }
    for (int si = 0; si < g.n_steps; ++si) {
        const Step &st = g.steps[si];
        if (st.kind != ST_WGRAD || st.m_lds < 0) continue;
        const int cnt = st.in * st.out + st.out;
        float *gm = st.net == 0 ? d.adam_m_actor : d.adam_m_critic, *gv = st.net == 0 ? d.adam_v_actor : d.adam_v_critic;
        for (int e = tid; e < cnt; e += kTrainThreads) {
            gm[st.w + e] = lds[st.m_lds + e];
            gv[st.w + e] = lds[st.m_lds + cnt + e];
        }
    }
    if (tid == 0) { d.adam_t[0] = tA; d.adam_t[1] = tC; }
}

// ---- host: LDS carve and the step list of one iteration -----------------------------------------------------------
struct NetDims {
    int in, h1, h2, out;     // actor: obs -> h1 -> h2 -> act ; critic: obs -> h1 (+act) -> h2 -> 1
    int extra;               // rows concatenated to the first hidden layer (critic: act_dim, actor: 0)
    int oW1() const { return 0; }
    int ob1() const { return in * h1; }
    int oW2() const { return ob1() + h1; }
    int ob2() const { return oW2() + (h1 + extra) * h2; }
    int oW3() const { return ob2() + h2; }
    int ob3() const { return oW3() + h2 * out; }
    int total() const { return ob3() + out; }
};

struct StepList {
    TrainArgs &g;
    bool overflow = false;
    explicit StepList(TrainArgs &a) : g(a) { g.n_steps = 0; }
    Step &add(int kind, bool barrier) {
        if (g.n_steps >= kMaxSteps) { overflow = true; g.n_steps = kMaxSteps - 1; }
        Step &s = g.steps[g.n_steps++];
        s = Step{};
        s.kind = (int16_t)kind; s.barrier = barrier ? 1 : 0; s.tail_from = 0x7fff; s.m_lds = -1;
        return s;
    }
    void fwd(int theta, const NetDims &n, int layer, int x, int z, int act, bool barrier, int xtail = -1, int tail_from = 0x7fff) {
        Step &s = add(ST_FWD, barrier);
        s.w = theta + (layer == 1 ? n.oW1() : layer == 2 ? n.oW2() : n.oW3());
        s.b = theta + (layer == 1 ? n.ob1() : layer == 2 ? n.ob2() : n.ob3());
        s.in = (int16_t)(layer == 1 ? n.in : layer == 2 ? n.h1 + n.extra : n.h2);
        s.out = (int16_t)(layer == 1 ? n.h1 : layer == 2 ? n.h2 : n.out);
        s.x = x; s.z = z; s.act = (int16_t)act;
        s.xtail = xtail >= 0 ? xtail : x; s.tail_from = (int16_t)tail_from;
    }
    // a_rows / act: activation rows whose derivative multiplies the result (act == ACT_NONE: none)
    void bwd(int theta_w, int out, int i0, int i1, int dz, int dx, int a_rows, int act, bool barrier) {
        Step &s = add(ST_BWD, barrier);
        s.w = theta_w; s.out = (int16_t)out; s.i0 = (int16_t)i0; s.i1 = (int16_t)i1; s.x = dz; s.z = dx;
        s.xtail = a_rows; s.act = (int16_t)act;
    }
    void deriv(int a, int dlt, int rows, int act, bool barrier) {
        Step &s = add(ST_DERIV, barrier);
        s.x = a; s.z = dlt; s.out = (int16_t)rows; s.act = (int16_t)act;
    }
    void wgrad(int net, int offW, int offb, int in, int out, int x, int dz, bool barrier, int xtail = -1, int tail_from = 0x7fff) {
        Step &s = add(ST_WGRAD, barrier);
        s.net = (int16_t)net; s.w = offW; s.b = offb; s.in = (int16_t)in; s.out = (int16_t)out; s.x = x; s.z = dz;
        s.xtail = xtail >= 0 ? xtail : x; s.tail_from = (int16_t)tail_from;
    }
};

}  // namespace ssc

using namespace ssc;

extern "C" size_t ssc_ddpg_train_workspace_bytes(const ssc_ddpg_desc *d) {
    if (d == nullptr || d->batch_size < 1 || d->batch_size > 4096 || d->obs_dim < 1 || d->act_dim < 1 || d->actor_h1 < 1 ||
        d->actor_h2 < 1 || d->critic_h1 < 1 || d->critic_h2 < 1)
        return 256;
    return ddpg_wide_workspace_bytes(d);
}

// Which kernel serves a shape: the shipped 64-32 / batch-64 shape has a kernel of its own (ddpg_train_fixed.hip); other
// nets <= 64 wide at batch 64 run the single-workgroup step interpreter below; everything else -- wider layers (the
// reference's 128-64 and 200-100 grid), other batch sizes -- the multi-workgroup kernels of ddpg_train_wide.hip, which
// need a workspace.  SSC_DDPG_INTERPRETER=1 / SSC_DDPG_WIDE=1 force a path (A/B measurements, and the tests that
// check every path against the oracle on the shipped shape).
static int ddpg_train_any(const char *who, const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx,
                          int32_t n_iters, float *d_losses, void *d_ws, size_t ws_bytes, bool have_ws, ssc_stream_t stream);

extern "C" int ssc_ddpg_train_ws(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                                 float *d_losses, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    return ddpg_train_any("ssc_ddpg_train_ws", d, rp, d_batch_idx, n_iters, d_losses, d_workspace, workspace_bytes, true, stream);
}

extern "C" int ssc_ddpg_train(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx,
                              int32_t n_iters, float *d_losses, ssc_stream_t stream) {
    return ddpg_train_any("ssc_ddpg_train", d, rp, d_batch_idx, n_iters, d_losses, nullptr, 0, false, stream);
}

static int ddpg_train_any(const char *who, const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx,
                          int32_t n_iters, float *d_losses, void *d_ws, size_t ws_bytes, bool have_ws, ssc_stream_t stream) {
    SSC_REQUIRE(d && rp, "%s: NULL descriptor", who);
    SSC_REQUIRE(n_iters >= 0, "%s: n_iters < 0", who);
    SSC_REQUIRE(d->batch_size >= 1, "%s: batch_size %d", who, d->batch_size);
    SSC_REQUIRE(d->obs_dim >= 1 && d->obs_dim <= SSC_MAX_STATE && d->act_dim >= 1 && d->act_dim <= SSC_MAX_ACT,
                "%s: obs_dim/act_dim out of range", who);
    SSC_REQUIRE(d->actor_h1 >= 1 && d->actor_h2 >= 1 && d->critic_h1 >= 1 && d->critic_h2 >= 1,
                "%s: bad hidden sizes", who);
    const char *force_i = getenv("SSC_DDPG_INTERPRETER"), *force_w = getenv("SSC_DDPG_WIDE");
    const bool want_interp = force_i && force_i[0] == '1', want_wide = force_w && force_w[0] == '1' && have_ws;
    SSC_REQUIRE(d->critic_l2_reg >= 0.0f, "%s: critic_l2_reg < 0", who);
    // (LayerNorm networks, critic_l2_reg and clip_norm run on the multi-workgroup kernels only)
    const bool narrow = !d->layer_norm && d->critic_l2_reg == 0.0f && !(d->clip_norm > 0.0f) && d->batch_size == kB && d->actor_h1 <= 64 && d->actor_h2 <= 64 && d->critic_h1 <= 64 && d->critic_h2 <= 64;
    if (!have_ws && !narrow)
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: batch_size %d / hidden layers wider than 64 / LayerNorm / critic_l2_reg / "
                                           "clip_norm run the multi-workgroup kernels, which need a workspace: call ssc_ddpg_train_ws", d->batch_size);
    if (n_iters == 0) return SSC_OK;
    SSC_REQUIRE(d->actor && d->critic && d->target_actor && d->target_critic && d->adam_m_actor && d->adam_v_actor &&
                    d->adam_m_critic && d->adam_v_critic && d->adam_t,
                "%s: NULL parameter / optimiser pointer", who);
    SSC_REQUIRE(rp->s && rp->a && rp->r && rp->t && rp->s2 && rp->capacity > 0 && d_batch_idx,
                "%s: NULL replay pointer", who);
    // the shipped 64-32 networks at a batch of several 64-row tiles: the straight-line kernel per tile, then the apply pass
    if (have_ws && !want_wide && !want_interp && ddpg_fixed_tiled_shape(d))
        return ddpg_train_fixed_tiled(d, rp, d_batch_idx, n_iters, d_losses, d_ws, ws_bytes, as_stream(stream));
    if (!narrow || want_wide) return ddpg_train_wide(d, rp, d_batch_idx, n_iters, d_losses, d_ws, ws_bytes, as_stream(stream));
    if (ddpg_fixed_shape(d) && !want_interp) return ddpg_train_fixed(d, rp, d_batch_idx, n_iters, d_losses, as_stream(stream));

    const NetDims A{d->obs_dim, d->actor_h1, d->actor_h2, d->act_dim, 0};
    const NetDims C{d->obs_dim, d->critic_h1, d->critic_h2, 1, d->act_dim};
    const int act2 = d->last_layer_tanh ? ACT_TANH : ACT_RELU;
    TrainArgs g{};
    g.d = *d; g.rp = *rp; g.batch_idx = d_batch_idx; g.n_iters = n_iters; g.losses = d_losses;
    // ---- LDS carve (rows of kP floats), then the parameters --------------------------------------------------
    int p = 0;
    auto take = [&](int rows) { const int q = p; p += rows * kP; return q; };
    const int S = take(d->obs_dim), RT = take(4);                             // RT: r, terminal, y (target Q), -Q(s, pi(s))
    const int X2 = take(C.h1 + d->act_dim);                                   // critic: relu(layer 1) rows, then the action rows
    const int CA2 = take(C.h2 > d->obs_dim ? C.h2 : d->obs_dim);              // critic layer 2; before that the next-state rows
    const int S2 = CA2;                                                       // (read by the target pass's first two steps only)
    const int DQ = take(1), DZ2 = take(C.h2);
    const int DZ1 = take(C.h1 + d->act_dim);                                  // critic layer-1 deltas; before that the target pass's X2B
    const int U1 = take(A.h1), U2 = take(A.h2), PI = take(d->act_dim);
    const int DZ3A = take(d->act_dim), DZ1A = take(A.h1);
    const int CB2 = take(C.h2 > A.h2 ? C.h2 : A.h2), DZB2 = take(C.h2);       // also target-pass scratch
    const int DZ2A = CB2;   // actor layer-2 deltas: written (bwd a3) after the last read of CB2, and in the target pass
                            // (target actor layer 2) read for the last time before CB2 is written
    const int X2B = DZ1;
    g.off_S = S; g.off_S2 = S2; g.off_RT = RT; g.off_X2act = X2 + C.h1 * kP;
    g.n_actor = A.total(); g.n_critic = C.total();
    // every parameter vector starts on a 16-byte boundary: the dense passes read weights as float4 (a
    // misaligned ds_read_b128 was measured 6x slower)
    auto al4 = [](int x) { return (x + 3) & ~3; };
    const int TA = al4(p), TC = al4(TA + A.total()), TTA = al4(TC + C.total()), TTC = al4(TTA + A.total());
    g.off_theta[0] = TA; g.off_theta[1] = TC; g.off_theta[2] = TTA; g.off_theta[3] = TTC;
    p = TTC + C.total();
    if ((size_t)p * sizeof(float) > 160 * 1024 - 128) { // 64 B of static LDS (loss partials, Adam step sizes)
        if (have_ws) return ddpg_train_wide(d, rp, d_batch_idx, n_iters, d_losses, d_ws, ws_bytes, as_stream(stream));
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: these layer sizes need %zu B of LDS on the single-workgroup path; "
                                           "call ssc_ddpg_train_ws", (size_t)p * sizeof(float));
    }
    // ---- the steps of one iteration ----------------------------------------------------------------------------
    StepList L(g);
    L.add(ST_GATHER, true);
    // target_Q = r + (1 - terminal) * gamma * Q'(s2, pi'(s2))      (ddpg_editted.py:132-133)
    L.fwd(TTA, A, 1, S2, DZ1A, ACT_RELU, false);
    L.fwd(TTC, C, 1, S2, X2B, ACT_RELU, true);
    L.fwd(TTA, A, 2, DZ1A, DZ2A, act2, true);
    L.fwd(TTA, A, 3, DZ2A, X2B + C.h1 * kP, ACT_TANH, true);
    L.fwd(TTC, C, 2, X2B, CB2, act2, true);
    L.fwd(TTC, C, 3, CB2, DZB2, ACT_NONE, true);
    g.steps[g.n_steps - 1].i0 = 1;                              // ... + the target_Q arithmetic on the row (post-op 1)
    // critic on (s, a) and actor on s                                (:181, :127)
    L.fwd(TC, C, 1, S, X2, ACT_RELU, false);
    L.fwd(TA, A, 1, S, U1, ACT_RELU, true);
    L.fwd(TC, C, 2, X2, CA2, act2, false);
    L.fwd(TA, A, 2, U1, U2, act2, true);
    L.fwd(TC, C, 3, CA2, DQ, ACT_NONE, false);
    g.steps[g.n_steps - 1].i0 = 2;                              // ... + critic loss, DQ becomes d loss / d q (post-op 2)
    L.fwd(TA, A, 3, U2, PI, ACT_TANH, true);
    // critic backward: dz2 = (W3 dq) * act'(z2)
    L.bwd(TC + C.oW3(), 1, 0, C.h2, DQ, DZ2, CA2, act2, false);
    // critic forward on (s, pi(s)) for the actor loss: the same relu(layer 1) rows, the action rows are pi(s)
    L.fwd(TC, C, 2, X2, CB2, act2, true, PI, C.h1);
    L.bwd(TC + C.oW2(), C.h2, 0, C.h1, DZ2, DZ1, X2, ACT_RELU, false);   // ... * relu'(layer 1)
    L.fwd(TC, C, 3, CB2, DZ3A, ACT_NONE, true);             // q(s, pi) -> DZ3A row 0, then
    g.steps[g.n_steps - 1].i0 = 3;                          // actor loss, the row becomes dq of -mean Q (post-op 3)
    L.bwd(TC + C.oW3(), 1, 0, C.h2, DZ3A, DZB2, CB2, act2, true);
    // d(-mean Q)/d(action) = rows h1.. of W2 dzb2 (through the output tanh), then back through the actor
    L.bwd(TC + C.oW2(), C.h2, C.h1, C.h1 + d->act_dim, DZB2, DZ3A, PI, ACT_TANH, true);
    L.bwd(TA + A.oW3(), A.out, 0, A.h2, DZ3A, DZ2A, U2, act2, true);
    L.bwd(TA + A.oW2(), A.h2, 0, A.h1, DZ2A, DZ1A, U1, ACT_RELU, false);
    L.add(ST_LOSSES, true);                                  // every read of the old parameters is done after this barrier
    L.wgrad(1, C.oW1(), C.ob1(), C.in, C.h1, S, DZ1, false);
    L.wgrad(1, C.oW2(), C.ob2(), C.h1 + d->act_dim, C.h2, X2, DZ2, false);
    L.wgrad(1, C.oW3(), C.ob3(), C.h2, 1, CA2, DQ, false);
    L.wgrad(0, A.oW1(), A.ob1(), A.in, A.h1, S, DZ1A, false);
    L.wgrad(0, A.oW2(), A.ob2(), A.h1, A.h2, U1, DZ2A, false);
    L.wgrad(0, A.oW3(), A.ob3(), A.h2, A.out, U2, DZ3A, true);
    L.add(ST_TUPDATE, true);
    if (L.overflow) return set_error(SSC_EINVAL, "ssc_ddpg_train: step list overflow");
    // Adam moments of the small layers move into what is left of the LDS (smallest layers first)
    {
        int free_floats = (160 * 1024 - 128) / 4 - p;
        for (int pass = 0; pass < 2; ++pass)
            for (int si = 0; si < g.n_steps; ++si) {
                Step &st = g.steps[si];
                if (st.kind != ST_WGRAD || st.m_lds >= 0) continue;
                const int cnt = st.in * st.out + st.out;
                if ((pass == 0 && cnt > 256) || cnt > 640 || 2 * cnt > free_floats) continue;
                st.m_lds = p;
                p += 2 * cnt;
                free_floats -= 2 * cnt;
            }
    }

    const size_t lds = (size_t)p * sizeof(float);
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(ddpg_train_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(ddpg_train_kernel)");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ddpg_train_kernel, dim3(1), dim3(kTrainThreads), lds, as_stream(stream), g);
    return check_launch("ssc_ddpg_train");
}
