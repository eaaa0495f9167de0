// mpc.hip -- SmartStart navigator MPC on the GPU: candidate action sampling, trajectory scoring
// and action selection (smartstart/RLAgents/NND_MB_agent.py:339-358, 498-520, 566-628) with the
// geometry helpers of smartstart/utilities/numerical.py:74-126.
//
// P independent problems (one per real env) x N candidate sequences; rows are problem-major.
//
// Scoring reproduces the reference INCLUDING its batch-global projection quirk: in
// projection_of_a_onto_b, np.sum(a*b) and np.sum(b*b) have no axis (numerical.py:89-92), so for
// every horizon step t the penalty of each sample depends on two scalars reduced over ALL N samples
// of the problem.  Hence two passes:
//   pass A  per sample: waypoint-index trajectory + per-t partial sums of a.b and b.b (f64)
//           -> deterministic per-block partials (no float atomics: bitwise reproducible)
//   pass B  per block: fixed-order reduction of the partials -> G1_t / G2_t; per sample: replay the
//           trajectory, add progress and penalty terms -> score; block argmax (lowest index wins);
//           the block of a problem that finishes last (atomic ticket) reduces the per-block winners.
// HBM traffic is the [H+1][P*N][d] trajectory read twice (8 B/sample-step for d=2) -- negligible
// next to the forward simulation that produced it.
#include "mpc_device.h"
#include "ssc_host.h"

namespace ssc {

constexpr int kMpcBlock = 256;
constexpr int kMaxH1 = 33;  // horizon + 1 <= 33

// State of one sample's walk over the horizon
struct WalkState {
    int idx;
    float score, prev, gpow;
    bool live;  // false: a lane past the last sample; it walks along (wave reductions) and contributes nothing
};

// One horizon step t of one sample at point pt.  PASS 0: accumulate a.b / b.b into sums[t][2].
// PASS 1: add the progress and penalty terms to the score using the global G1/G2 of step t.
// G = lanes that share one problem's reduction (64: a whole wave; 16 / 32: the small-N kernel's sub-wave groups).
// DEFER (pass A, prefetched trajectories): the lane's a.b / b.b of step t go to LDS as they are (acc[t][2][thread]) and the
// block sums them once, after the walk (block_sums_deferred) -- the per-step fp64 butterflies were 24 ds_bpermute + 12 v_add_f64
// per horizon step and wave, more than the walk itself.
template <int PASS, int G = 64, bool DEFER = false>
__device__ __forceinline__ void walk_step(const MpcArgs &a, const float *pt, int t, const float *wps, const float *lefts,
                                          int W, const float *inv_r, double *wave_sums, const float *cproj,
                                          WalkState &w, float *acc = nullptr) {
    const int d = a.d;
    if (t == 0) w.prev = lefts[w.idx] + ell_dist(pt, wps + w.idx * d, inv_r, d);  // NND_MB_agent.py:573-576
    const int nxt = min(w.idx + 1, W - 1);
    float dc = ell_dist(wps + w.idx * d, pt, inv_r, d);   // :589
    const float dn = ell_dist(wps + nxt * d, pt, inv_r, d);  // :590
    const bool move = (dc <= a.theta || dn <= dc) && w.idx != W - 1;  // move_to_next :491-496
    w.idx += move ? 1 : 0;                                // :600
    dc = move ? dn : dc;                                  // :603
    const float end = lefts[w.idx] + dc;                  // :606
    if (PASS == 1) w.score += (w.prev - end) * w.gpow;    // :609
    w.prev = end;                                         // :610
    const int b = max(w.idx - 1, 0);                      // :615
    // dist_line_seg_to_point (numerical.py:74-81) in radii-scaled coordinates
    float ab = 0.0f, bb = 0.0f;
    float av[SSC_MAX_STATE], bv[SSC_MAX_STATE];
#pragma unroll
    for (int k = 0; k < SSC_MAX_STATE; ++k)
        if (k < d) {
            av[k] = (pt[k] - wps[b * d + k]) * inv_r[k];
            bv[k] = (wps[(b + 1) * d + k] - wps[b * d + k]) * inv_r[k];
            ab = fmaf(av[k], bv[k], ab);
            bb = fmaf(bv[k], bv[k], bb);
        }
    if (PASS == 0 && DEFER) {
        acc[(t * 2 + 0) * kMpcBlock + threadIdx.x] = w.live ? ab : 0.0f;
        acc[(t * 2 + 1) * kMpcBlock + threadIdx.x] = w.live ? bb : 0.0f;
    } else if (PASS == 0) {
        // every lane of the wave is at the same t: reduce now (fixed butterfly order) instead of keeping a
        // per-thread [H+1][2] array of doubles, which a runtime t would push into scratch memory
        double v0 = w.live ? (double)ab : 0.0, v1 = w.live ? (double)bb : 0.0;
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) {
            v0 += __shfl_xor(v0, m);
            v1 += __shfl_xor(v1, m);
        }
        if ((threadIdx.x & (G - 1)) == 0) {
            wave_sums[t * 2 + 0] = v0;
            wave_sums[t * 2 + 1] = v1;
        }
    } else {
        // proj = (sum(a.b) / sum(b.b)) * b ; distance(proj, a)   (numerical.py:84-98 + :116-124)
        const float c = a.per_row ? ab / bb : cproj[t];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < SSC_MAX_STATE; ++k)
            if (k < d) {
                const float v = fmaf(c, bv[k], -av[k]);
                s = fmaf(v, v, s);
            }
        w.score -= sqrtf(s) * a.hpf * a.gamma;  // :622 (gamma, not gamma^t)
        w.gpow *= a.gamma;
    }
}

// One sample's walk, trajectory points read from S step by step (any horizon / state dim)
template <int PASS, int G = 64>
__device__ __forceinline__ float mpc_walk(const MpcArgs &a, const float *__restrict__ S, int64_t row, int64_t M,
                                          const float *wps, const float *lefts, int W, int idx0,
                                          const float *inv_r, bool live, double *wave_sums, const float *cproj) {
    const int d = a.d;
    float pt[SSC_MAX_STATE];
    WalkState w{idx0, 0.0f, 0.0f, 1.0f, live};
    for (int t = 0; t <= a.H; ++t) {
        const float *p = S + ((int64_t)t * M + row) * d;
#pragma unroll
        for (int k = 0; k < SSC_MAX_STATE; ++k) pt[k] = (live && k < d) ? p[k] : 0.0f;
        walk_step<PASS, G>(a, pt, t, wps, lefts, W, inv_r, wave_sums, cproj, w);
    }
    return w.score;
}

// The same walk with the whole trajectory of the sample fetched up front (H + 1 <= kPreT, d <= kPreD): all
// loads are in flight together, one memory latency per sample instead of one per horizon step.
constexpr int kPreT = 8, kPreD = 4;
struct PrePts {
    float v[kPreT][SSC_MAX_STATE];
};
// Issued FIRST in both passes, before the problem's waypoints are staged and (pass B) the partial sums are reduced: the
// three are independent round trips to L2 / HBM, and these small launches are nothing but a chain of such round trips.
__device__ __forceinline__ void mpc_fetch_pts(const MpcArgs &a, const float *__restrict__ S, int64_t row, int64_t M, bool live,
                                              PrePts &pts) {
#pragma unroll
    for (int t = 0; t < kPreT; ++t)
#pragma unroll
        for (int k = 0; k < SSC_MAX_STATE; ++k)
            pts.v[t][k] = (live && k < kPreD && t <= a.H && k < a.d) ? S[((int64_t)t * M + row) * a.d + k] : 0.0f;
}

template <int PASS, int G = 64, bool DEFER = false>
__device__ __forceinline__ float mpc_walk_pre(const MpcArgs &a, const PrePts &pp, const float *wps, const float *lefts, int W,
                                              int idx0, const float *inv_r, bool live, double *wave_sums, const float *cproj,
                                              float *acc = nullptr) {
    const float (&pts)[kPreT][SSC_MAX_STATE] = pp.v;
    WalkState w{idx0, 0.0f, 0.0f, 1.0f, live};
#pragma unroll
    for (int t = 0; t < kPreT; ++t)
        if (t <= a.H) walk_step<PASS, G, DEFER>(a, pts[t], t, wps, lefts, W, inv_r, wave_sums, cproj, w, acc);
    return w.score;
}

// The block's sums of the deferred terms: sum e = (t, a.b | b.b) belongs to the 16 lanes of group e; a lane adds threads
// lane, lane + 16, ... in order (fp64), the group's 16 subtotals meet in a butterfly -- the same order in every run.
__device__ __forceinline__ void block_sums_deferred(const float *acc, int n_sums, double *out) {
    const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    double v = 0.0;
    if (grp < n_sums) {
        const float *src = acc + grp * kMpcBlock + l16;
#pragma unroll
        for (int j = 0; j < kMpcBlock / 16; ++j) v += (double)src[16 * j];
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (l16 == 0 && grp < n_sums) out[grp] = v;
}

__device__ __forceinline__ bool mpc_can_prefetch(const MpcArgs &a) { return a.H + 1 <= kPreT && a.d <= kPreD; }  // block-uniform

template <int PASS, int G = 64>
__device__ __forceinline__ float mpc_walk_any(const MpcArgs &a, const PrePts &pp, const float *__restrict__ S, int64_t row,
                                              int64_t M, const float *wps, const float *lefts, int W, int idx0,
                                              const float *inv_r, bool live, double *wave_sums, const float *cproj) {
    if (mpc_can_prefetch(a)) return mpc_walk_pre<PASS, G>(a, pp, wps, lefts, W, idx0, inv_r, live, wave_sums, cproj);
    return mpc_walk<PASS, G>(a, S, row, M, wps, lefts, W, idx0, inv_r, live, wave_sums, cproj);
}

// The walk of a sample that starts at waypoint idx0 can only touch waypoints idx0-1 ... idx0+H+1 (the index advances
// by at most one per horizon step, NND_MB_agent.py:600; the penalty segment starts at max(idx-1, 0), :615), so a
// WINDOW of at most H + 4 waypoints is staged -- not the whole plan (up to ~1000 waypoints per problem in the
// reference's runs, which the first versions of this file copied into 48 KB of LDS per block).  The walk then runs in
// window-relative indices: W and idx0 come back relative to the window base max(idx0-1, 0), which leaves every
// comparison of the walk (idx != W-1, min(idx+1, W-1), max(idx-1, 0)) unchanged.
constexpr int kWinMax = kMaxH1 + 3;                   // waypoints in a window
constexpr int kWinFloats = kWinMax * (SSC_MAX_STATE + 1) + SSC_MAX_STATE;   // wps | lefts | inv_r

// lanes `lane` of `nl` cooperating lanes stage problem p's window; the caller synchronises them afterwards
// (q = plan_index(a, p) and cur = a.cur_idx[p] come in as values: a caller that needs other per-problem data too can request
// them all in one round trip)
__device__ __forceinline__ void load_window_at(const MpcArgs &a, int q, int cur, int lane, int nl, float *win, int &W, int &idx0) {
    const int off = a.wp_off[q];
    const int Wabs = plan_len(a, q);
    // (clamped: a plan slot re-published under an env that still follows it -- a pool too small for the refresh cadence --
    // may be shorter than the env's waypoint index; the env then heads for the new plan's last waypoint instead of reading
    // outside the window)
    const int iabs = min(cur, max(Wabs - 1, 0));
    const int wb = max(iabs - 1, 0);
    const int cnt = max(min(Wabs - wb, a.H + 4), 0);
    float *wps = win, *lefts = win + kWinMax * a.d, *inv_r = lefts + kWinMax;
    const float *wsrc = a.wp + (int64_t)(off + wb) * a.d;
    if (cnt * a.d <= nl) {   // one element of each array per lane: the three loads together (clamped index), then the stores
        const float wv = wsrc[max(min(lane, cnt * a.d - 1), 0)], lv = a.left[off + wb + max(min(lane, cnt - 1), 0)];
        const float rv = a.radii[q * a.d + min(lane, a.d - 1)];
        if (lane < cnt * a.d) wps[lane] = wv;
        if (lane < cnt) lefts[lane] = lv;
        if (lane < a.d) inv_r[lane] = 1.0f / rv;
    } else {
        for (int e = lane; e < cnt * a.d; e += nl) wps[e] = wsrc[e];
        for (int e = lane; e < cnt; e += nl) lefts[e] = a.left[off + wb + e];
        if (lane < a.d) inv_r[lane] = 1.0f / a.radii[q * a.d + lane];
    }
    W = Wabs - wb;
    idx0 = iabs - wb;
}
__device__ __forceinline__ void load_window(const MpcArgs &a, int p, int lane, int nl, float *win, int &W, int &idx0) {
    load_window_at(a, plan_index(a, p), a.cur_idx[p], lane, nl, win, W, idx0);   // plan_index == p unless the problems share a plan pool
}

// The same staging in two phases for a whole block (nl = kMpcBlock >= 36 * 8 / 2): `issue` computes the window and puts its
// loads in flight (<= 2 waypoint floats, one distance, one radius per thread), `commit` writes them to LDS -- so that a
// caller can request other data between the two and pay ONE round trip for both.
struct WindowLoads {
    float w0, w1, l0, r0;
    int cnt, d;
};
__device__ __forceinline__ WindowLoads load_window_issue(const MpcArgs &a, int p, int &W, int &idx0) {
    const int q = plan_index(a, p);
    const int off = a.wp_off[q];
    const int Wabs = plan_len(a, q);
    const int iabs = min(a.cur_idx[p], max(Wabs - 1, 0));     // (clamped: see load_window)
    const int wb = max(iabs - 1, 0);
    WindowLoads r;
    r.cnt = max(min(Wabs - wb, a.H + 4), 0);
    r.d = a.d;
    const float *wsrc = a.wp + (int64_t)(off + wb) * a.d;
    const int t = threadIdx.x, n = r.cnt * a.d;
    r.w0 = t < n ? wsrc[t] : 0.0f;
    r.w1 = t + kMpcBlock < n ? wsrc[t + kMpcBlock] : 0.0f;
    r.l0 = t < r.cnt ? a.left[off + wb + t] : 0.0f;
    r.r0 = t < a.d ? a.radii[q * a.d + t] : 1.0f;
    W = Wabs - wb;
    idx0 = iabs - wb;
    return r;
}
__device__ __forceinline__ void load_window_commit(const WindowLoads &r, float *win) {
    float *wps = win, *lefts = win + kWinMax * r.d, *inv_r = lefts + kWinMax;
    const int t = threadIdx.x, n = r.cnt * r.d;
    if (t < n) wps[t] = r.w0;
    if (t + kMpcBlock < n) wps[t + kMpcBlock] = r.w1;
    if (t < r.cnt) lefts[t] = r.l0;
    if (t < r.d) inv_r[t] = 1.0f / r.r0;
}
static_assert(kWinMax * SSC_MAX_STATE <= 2 * 256, "two waypoint floats per thread cover a window");

__device__ __forceinline__ void load_problem(const MpcArgs &a, int p, float *win, int &W, int &idx0) {
    load_window(a, p, threadIdx.x, blockDim.x, win, W, idx0);
    __syncthreads();
}

// argmax over the block, lowest index on ties (np.argmax, NND_MB_agent.py:626); result valid in thread 0
__device__ __forceinline__ void block_argmax(float &score, int &best, float *rs, int *ri) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float os = __shfl_xor(score, m);
        const int oi = __shfl_xor(best, m);
        if (os > score || (os == score && oi < best)) { score = os; best = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { rs[wave] = score; ri[wave] = best; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < kMpcBlock / 64; ++w)
            if (rs[w] > score || (rs[w] == score && ri[w] < best)) { score = rs[w]; best = ri[w]; }
}

// The kernels below are compiled per state dimension (D = 0: any d, read from the arguments) and run a body in which `a.d` is
// a known constant -- separate kernels, not branches of one (a kernel's register count is that of its widest branch, and
// the any-d copy keeps 64 prefetched floats per sample): the walk's distance and projection loops are written over SSC_MAX_STATE = 8 slots with `k < d` guards,
// which for the 2- and 3-d states of the shipped envs is otherwise two thirds predicated-off work per sample and point.
#define SSC_MPC_LAUNCH_D(d, KERNEL_OF_D, ...)                                   \
    do {                                                                        \
        if ((d) == 2) hipLaunchKernelGGL((KERNEL_OF_D(2)), __VA_ARGS__);       \
        else if ((d) == 3) hipLaunchKernelGGL((KERNEL_OF_D(3)), __VA_ARGS__);  \
        else if ((d) == 1) hipLaunchKernelGGL((KERNEL_OF_D(1)), __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL_OF_D(0)), __VA_ARGS__);                \
    } while (0)

// Blocks are numbered problem-major in grid.x (block = p * nblk + b): a problem count of 65 536 -- one navigation
// problem per env at the BASELINE env count -- does not fit gridDim.y.
__device__ __forceinline__ void mpc_pass_a_body(MpcArgs a, const float *__restrict__ S, double *__restrict__ partial,
                                                int32_t *__restrict__ ticket) {
    __shared__ double red[4 * kMaxH1 * 2];  // [4 waves][H+1][2]
    __shared__ float win[kWinFloats];
    __shared__ float acc[kPreT * 2 * kMpcBlock];   // deferred sums of the prefetched walk: [t][a.b | b.b][thread]
    static_assert(kPreT * 2 <= kMpcBlock / 16, "one 16-lane group per deferred sum");
    const int p = blockIdx.x / a.nblk, bx = blockIdx.x - p * a.nblk;
    if (bx == 0 && threadIdx.x == 0) ticket[p] = 0;  // pass B elects its last block with it
    const int n = bx * kMpcBlock + threadIdx.x;
    const bool live = n < a.N;
    const int64_t row = (int64_t)p * a.N + (live ? n : 0), M = (int64_t)a.P * a.N;
    PrePts pp;
    if (mpc_can_prefetch(a)) mpc_fetch_pts(a, S, row, M, live, pp);
    int W, idx0;
    load_problem(a, p, win, W, idx0);
    const float *wps = win, *lefts = win + kWinMax * a.d, *inv_r = lefts + kWinMax;
    if (mpc_can_prefetch(a)) {    // block-uniform
        mpc_walk_pre<0, 64, true>(a, pp, wps, lefts, W, idx0, inv_r, live, nullptr, nullptr, acc);
        __syncthreads();
        block_sums_deferred(acc, (a.H + 1) * 2, partial + ((int64_t)p * a.nblk + bx) * (a.H + 1) * 2);
        return;
    }
    mpc_walk<0>(a, S, row, M, wps, lefts, W, idx0, inv_r, live, red + (threadIdx.x >> 6) * kMaxH1 * 2, nullptr);
    __syncthreads();
    for (int e = threadIdx.x; e < (a.H + 1) * 2; e += kMpcBlock) {
        double v = 0.0;
        for (int w = 0; w < kMpcBlock / 64; ++w) v += red[w * kMaxH1 * 2 + e];
        partial[((int64_t)p * a.nblk + bx) * (a.H + 1) * 2 + e] = v;
    }
}

template <int D>
__global__ __launch_bounds__(kMpcBlock) void mpc_pass_a_kernel(MpcArgs a, const float *__restrict__ S,
                                                               double *__restrict__ partial,
                                                               int32_t *__restrict__ ticket) {
    if (D > 0) a.d = D;
    mpc_pass_a_body(a, S, partial, ticket);
}

// What ssc_mpc_select_action does, folded into pass B's last block (ssc_mpc_score_select)
struct SelectArgs {
    float *action;       // [P][act]; nullptr: no selection epilogue
    float *best_path;    // [P][H+1][d] or nullptr
    const float *A;      // [P*N][H][act] or nullptr -> regenerate the winner's first action from the sampling spec
    int32_t act;
    float noise;
    uint64_t seed, pid0, t;          // noise key (ssc_mpc_select_action)
    uint64_t s_seed, s_pid0, s_t;    // sampling spec (ssc_mpc_sample_actions)
    const uint64_t *t_base;          // device step counter added to t and s_t
    float low[SSC_MAX_ACT], span[SSC_MAX_ACT];
};

// get_action_with_predicted_states tail (NND_MB_agent.py:339-358) for problem p whose winner is sample `best`, by
// lanes `lane` of `nl` cooperating lanes; the same draws as mpc_select_kernel.
__device__ __forceinline__ void mpc_select_epilogue(const MpcArgs &a, const SelectArgs &sel, const float *__restrict__ S,
                                                    int p, int best, int lane, int nl) {
    const uint64_t tb = sel.t_base != nullptr ? *sel.t_base : 0;
    if (lane < sel.act) {
        const int ai = lane;
        float first;
        if (sel.A != nullptr) {
            first = sel.A[((int64_t)p * a.N + best) * a.H * sel.act + ai];
        } else {   // flat index h * act + ai = ai < 4: word ai of the sample's first Philox call
            const int per = (a.H * sel.act + 3) / 4;
            const u32x4 w = rng_words(sel.s_seed, ((sel.s_pid0 + (uint64_t)p) << 32) + (uint64_t)best,
                                      (sel.s_t + tb) * (uint64_t)per, TAG_MPC);
            first = uniform_f32(pick(w, (uint32_t)ai), sel.low[ai], sel.span[ai]);
        }
        float g = 0.0f;
        if (sel.noise != 0.0f) {
            const u32x4 w = rng_words(sel.seed, sel.pid0 + (uint64_t)p, sel.t + tb, TAG_MPC_NOISE);
            const u32x4 w2 = rng_words(sel.seed, sel.pid0 + (uint64_t)p, sel.t + tb + ((uint64_t)1 << 40), TAG_MPC_NOISE);
            g = (ai == 0) ? gaussian_f32(w.x, w.y) : (ai == 1) ? gaussian_f32(w.z, w.w)
                : (ai == 2) ? gaussian_f32(w2.x, w2.y) : gaussian_f32(w2.z, w2.w);
        }
        sel.action[p * sel.act + ai] = first + sel.noise * g;
    }
    if (sel.best_path != nullptr) {
        const int64_t row = (int64_t)p * a.N + best, M = (int64_t)a.P * a.N;
        for (int e = lane; e < (a.H + 1) * a.d; e += nl) {
            const int tt = e / a.d, k = e % a.d;
            sel.best_path[(int64_t)p * (a.H + 1) * a.d + e] = S[((int64_t)tt * M + row) * a.d + k];
        }
    }
}

__device__ __forceinline__ void mpc_pass_b_body(MpcArgs a, int stage_partials, const SelectArgs &sel, const float *__restrict__ S,
                                                const double *__restrict__ partial, float *__restrict__ scores,
                                                float *blk_best_score, int32_t *blk_best_idx, int32_t *__restrict__ ticket,
                                                int32_t *__restrict__ best_idx, float *__restrict__ best_score) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [nblk][H+1][2] doubles when stage_partials
    __shared__ float cproj[kMaxH1 + 1];   // [H+1] sum(a.b) / sum(b.b) of the whole problem
    __shared__ float rs[4];               // block argmax scratch
    __shared__ int ri[4], last[2];
    __shared__ float win[kWinFloats];
    double *pstage = stage_partials ? reinterpret_cast<double *>(smem) : nullptr;
    const int p = blockIdx.x / a.nblk, bx = blockIdx.x - p * a.nblk;
    const int n = bx * kMpcBlock + threadIdx.x;
    const int64_t M = (int64_t)a.P * a.N;
    PrePts pp;
    if (mpc_can_prefetch(a)) mpc_fetch_pts(a, S, (int64_t)p * a.N + (n < a.N ? n : 0), M, n < a.N, pp);
    // fixed-order reduction of the per-block partials of this problem.  The partials come in with ONE round trip (every
    // thread fetches its share into LDS) and are then summed in block order by thread t -- H + 1 threads walking nblk
    // dependent global loads each was most of this kernel's run time.
    // The partials, the waypoint window (behind its index loads) and the trajectory points are requested TOGETHER: one
    // wait, one barrier -- with the window loaded after the reduction every block paid one more dependent round trip
    // (five in all, at ~7 blocks per CU the whole launch is a chain of them).
    const int n_part = a.nblk * (a.H + 1) * 2;
    const bool one_trip = pstage != nullptr && n_part <= kMpcBlock;     // block-uniform
    double pv = 0.0;
    if (one_trip && threadIdx.x < n_part) pv = partial[(int64_t)p * n_part + threadIdx.x];
    int W, idx0;
    const WindowLoads wl = load_window_issue(a, p, W, idx0);
    if (one_trip) {
        if (threadIdx.x < n_part) pstage[threadIdx.x] = pv;
    } else if (pstage != nullptr) {
        for (int e = threadIdx.x; e < n_part; e += kMpcBlock) pstage[e] = partial[(int64_t)p * n_part + e];
    }
    load_window_commit(wl, win);
    __syncthreads();
    for (int t = threadIdx.x; t <= a.H; t += kMpcBlock) {
        double g0 = 0.0, g1 = 0.0;
        for (int b = 0; b < a.nblk; ++b) {
            const int e = (b * (a.H + 1) + t) * 2;
            g0 += pstage != nullptr ? pstage[e + 0] : partial[(int64_t)p * n_part + e + 0];
            g1 += pstage != nullptr ? pstage[e + 1] : partial[(int64_t)p * n_part + e + 1];
        }
        cproj[t] = (float)(g0 / g1);
    }
    __syncthreads();
    const float *wps = win, *lefts = win + kWinMax * a.d, *inv_r = lefts + kWinMax;
    float score = -INFINITY;
    int best = 0x7fffffff;
    if (n < a.N) {
        score = mpc_walk_any<1>(a, pp, S, (int64_t)p * a.N + n, M, wps, lefts, W, idx0, inv_r, true, nullptr, cproj);
        scores[(int64_t)p * a.N + n] = score;
        best = n;
        if (isnan(score)) score = -INFINITY;  // np.argmax would return the first NaN; we skip NaNs
    }
    block_argmax(score, best, rs, ri);
    // the block that finishes last reduces the per-block winners of its problem (no third launch)
    if (threadIdx.x == 0) {
        // write-through (sc1) stores, drained by THIS lane, then a relaxed agent-scope ticket: the last block reads the
        // winners back with sc1 loads after its own add returned (MI355X_MICROARCH.md "Valid forms", first table row).
        // An acq_rel ticket would add an L2 write-back + invalidate (~3.5 us) to every block's critical path.  This
        // library is built for gfx950 only (csrc/Makefile: --offload-arch=gfx950); the handoff relies on that
        // target's agent-scope atomics being coherent at the memory side, see the static_assert below.
        __hip_atomic_store(&blk_best_score[p * a.nblk + bx], score, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&blk_best_idx[p * a.nblk + bx], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int tk = __hip_atomic_fetch_add(&ticket[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last[0] = (tk == a.nblk - 1);
    }
    __syncthreads();
    if (!last[0]) return;
    score = -INFINITY;
    best = 0x7fffffff;
    for (int b = threadIdx.x; b < a.nblk; b += kMpcBlock) {
        const float os = __hip_atomic_load(&blk_best_score[p * a.nblk + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int oi = __hip_atomic_load(&blk_best_idx[p * a.nblk + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (os > score || (os == score && oi < best)) { score = os; best = oi; }
    }
    __syncthreads();  // rs/ri are reused
    block_argmax(score, best, rs, ri);
    if (threadIdx.x == 0) {
        best = (best == 0x7fffffff) ? 0 : best;
        best_idx[p] = best;
        if (best_score) best_score[p] = score;
        ri[0] = best;
    }
    if (sel.action == nullptr) return;
    __syncthreads();
    mpc_select_epilogue(a, sel, S, p, ri[0], threadIdx.x, kMpcBlock);
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mpc.hip: the relaxed ticket handoff of pass B is written for gfx950 (MI355X) only"
#endif

template <int D>
__global__ __launch_bounds__(kMpcBlock) void mpc_pass_b_kernel(MpcArgs a, int stage_partials, SelectArgs sel, const float *__restrict__ S,
                                                               const double *__restrict__ partial,
                                                               float *__restrict__ scores,
                                                               float *blk_best_score, int32_t *blk_best_idx,
                                                               int32_t *__restrict__ ticket,
                                                               int32_t *__restrict__ best_idx,
                                                               float *__restrict__ best_score) {
    if (D > 0) a.d = D;
    mpc_pass_b_body(a, stage_partials, sel, S, partial, scores, blk_best_score, blk_best_idx, ticket, best_idx, best_score);
}

// ---- N <= 64 samples per problem: the whole scoring of a problem in ONE launch ---------------------------------------
// With all samples of a problem inside one wave the two batch-global sums of numerical.py:89-92 are a butterfly over
// the G = 16 / 32 / 64 lanes that hold the problem, so nothing crosses a block: pass A, the reduction, pass B, the
// argmax and the selection epilogue run back to back in one kernel, 256 / G problems per block.  This is the shape of
// "one navigator per env" at the BASELINE env count (P = 65 536 problems x 16-64 candidates); the f64 sums are formed
// in the same order as the two-pass path forms them for N <= 64 (lanes past N add exact zeros), so scores and
// winners are bit-identical to it.
template <int G, int D>
__device__ __forceinline__ void mpc_small_body(MpcArgs a, const SelectArgs &sel, const float *__restrict__ S,
                                               float *__restrict__ scores, int32_t *__restrict__ best_idx,
                                               float *__restrict__ best_score) {
    constexpr int kGroups = kMpcBlock / G;
    constexpr int DM = D > 0 ? D : SSC_MAX_STATE;      // a window is wps [kWinMax][d] | lefts [kWinMax] | inv_r [d]
    __shared__ float win[kGroups][kWinMax * (DM + 1) + DM];
    __shared__ double sums[kGroups][kMaxH1 * 2];
    __shared__ float cproj[kGroups][kMaxH1 + 1];
    const int g = threadIdx.x / G, n = threadIdx.x & (G - 1);
    const int pq = blockIdx.x * kGroups + g;
    int n_serve = a.P;
    if (a.live_list != nullptr) {   // compact work list: group pq serves problem live_list[pq]
        n_serve = *a.n_live;
        if (blockIdx.x * kGroups >= n_serve) return;   // block-uniform, in front of every barrier
    }
    const bool has = pq < n_serve;
    const int ps = has ? pq : n_serve - 1;       // a group past the last problem shadows it and writes nothing
    const int p = a.live_list != nullptr ? a.live_list[ps] : ps;
    // the mask byte, the plan index and the waypoint index need nothing but p: one round trip for the three
    const int q_pre = plan_index(a, p), cur_pre = a.cur_idx[p];
    // a problem masked out by a.active (an env that is not navigating) costs its group three barriers and nothing else
    const bool scored = a.active == nullptr || a.active[p] != 0;    // group-uniform
    const bool live = has && n < a.N && scored;
    const int64_t row = (int64_t)p * a.N + (n < a.N ? n : 0), M = (int64_t)a.P * a.N;
    PrePts pp;
    if (scored && mpc_can_prefetch(a)) mpc_fetch_pts(a, S, row, M, live, pp);
    int W = 2, idx0 = 0;
    if (scored) load_window_at(a, q_pre, cur_pre, n, G, win[g], W, idx0);
    __syncthreads();
    const float *wps = win[g], *lefts = win[g] + kWinMax * a.d, *inv_r = lefts + kWinMax;
    if (scored) mpc_walk_any<0, G>(a, pp, S, row, M, wps, lefts, W, idx0, inv_r, live, sums[g], nullptr);
    __syncthreads();
    if (scored)
        for (int t = n; t <= a.H; t += G) cproj[g][t] = (float)(sums[g][t * 2 + 0] / sums[g][t * 2 + 1]);
    __syncthreads();
    if (!scored) return;     // no barrier below
    float score = -INFINITY;
    int best = 0x7fffffff;
    if (n < a.N) {   // (a shadow group walks too: uniform control flow, no store)
        score = mpc_walk_any<1, G>(a, pp, S, row, M, wps, lefts, W, idx0, inv_r, true, nullptr, cproj[g]);
        if (has) scores[(int64_t)p * a.N + n] = score;
        best = n;
        if (isnan(score)) score = -INFINITY;
    }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) {
        const float os = __shfl_xor(score, m);
        const int oi = __shfl_xor(best, m);
        if (os > score || (os == score && oi < best)) { score = os; best = oi; }
    }
    best = (best == 0x7fffffff) ? 0 : best;       // every lane of the group holds the winner now
    if (!has) return;
    if (n == 0) {
        best_idx[p] = best;
        if (best_score) best_score[p] = score;
    }
    if (sel.action != nullptr) mpc_select_epilogue(a, sel, S, p, best, n, G);
}

template <int G, int D>
__global__ __launch_bounds__(kMpcBlock) void mpc_small_kernel(MpcArgs a, SelectArgs sel, const float *__restrict__ S,
                                                              float *__restrict__ scores, int32_t *__restrict__ best_idx,
                                                              float *__restrict__ best_score) {
    if (D > 0) a.d = D;
    mpc_small_body<G, D>(a, sel, S, scores, best_idx, best_score);
}

// The navigating envs as a compact list (ssc_nav_compact): ballot + prefix inside a wave, the 16 wave counts of a block
// scanned in LDS, ONE atomic per 1024 envs for the block's base (an atomic per wave -- 1024 of them on one address at
// 65 536 envs -- made this a 10 us kernel; 64 of them make it a 2 us one).
constexpr int kCompactBlock = 1024;
__global__ __launch_bounds__(kCompactBlock) void nav_compact_kernel(int64_t n, const uint8_t *__restrict__ mode, int32_t *__restrict__ list,
                                                                    int32_t *__restrict__ count) {
    __shared__ int32_t wave_cnt[kCompactBlock / 64];
    __shared__ int32_t block_base;
    const int64_t i = (int64_t)blockIdx.x * kCompactBlock + threadIdx.x;
    const bool nav = i < n && mode[i] != 0;
    const uint64_t b = __builtin_amdgcn_ballot_w64(nav);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __builtin_popcountll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t total = 0;
#pragma unroll
        for (int w = 0; w < kCompactBlock / 64; ++w) {
            const int32_t c = wave_cnt[w];
            wave_cnt[w] = total;      // exclusive prefix
            total += c;
        }
        block_base = total > 0 ? atomicAdd(count, total) : 0;
    }
    __syncthreads();
    if (nav) list[block_base + wave_cnt[wave] + __builtin_popcountll(b & ((1ull << lane) - 1ull))] = (int32_t)i;
}

struct ActBounds {
    float low[SSC_MAX_ACT], span[SSC_MAX_ACT];
};

// npr.uniform(low, high, (N, H, act)) (NND_MB_agent.py:500-501); oracle: mpc_action_samples
__global__ __launch_bounds__(256) void mpc_sample_kernel(int P, int N, int H, int act, ActBounds bd, uint64_t seed,
                                                         uint64_t problem_id0, uint64_t t, const uint64_t *__restrict__ t_base,
                                                         float *__restrict__ A) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= (int64_t)P * N) return;
    if (t_base != nullptr) t += *t_base;   // device-resident step counter (HIP-graph replay)
    const uint64_t p = (uint64_t)(row / N), n = (uint64_t)(row % N);
    const uint64_t id = ((problem_id0 + p) << 32) + n;
    const int per = (H * act + 3) / 4;
    float *out = A + row * H * act;
    for (int c = 0; c < per; ++c) {
        const u32x4 w = rng_words(seed, id, t * (uint64_t)per + (uint64_t)c, TAG_MPC);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = c * 4 + j;
            if (f < H * act) {
                const int ai = f % act;
                out[f] = uniform_f32(pick(w, j), bd.low[ai], bd.span[ai]);
            }
        }
    }
}

// action = best_sequence[0] + noise_amount * N(0,1), no clip (NND_MB_agent.py:353-356); best_path (:518)
__global__ __launch_bounds__(64) void mpc_select_kernel(int P, int N, int H, int d, int act,
                                                        const float *__restrict__ A, const float *__restrict__ S,
                                                        const int32_t *__restrict__ best_idx, float noise,
                                                        uint64_t seed, uint64_t problem_id0, uint64_t t,
                                                        float *__restrict__ action, float *__restrict__ best_path) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    const int64_t row = (int64_t)p * N + best_idx[p];
    const u32x4 w = rng_words(seed, problem_id0 + (uint64_t)p, t, TAG_MPC_NOISE);
    const u32x4 w2 = rng_words(seed, problem_id0 + (uint64_t)p, t + ((uint64_t)1 << 40), TAG_MPC_NOISE);
    for (int a = 0; a < act; ++a) {
        float g = 0.0f;
        if (noise != 0.0f) {
            g = (a == 0) ? gaussian_f32(w.x, w.y) : (a == 1) ? gaussian_f32(w.z, w.w)
                : (a == 2) ? gaussian_f32(w2.x, w2.y) : gaussian_f32(w2.z, w2.w);
        }
        action[p * act + a] = A[row * H * act + a] + noise * g;
    }
    if (best_path != nullptr)
        for (int tt = 0; tt <= H; ++tt)
            for (int k = 0; k < d; ++k)
                best_path[((int64_t)p * (H + 1) + tt) * d + k] = S[((int64_t)tt * P * N + row) * d + k];
}

// NND_MB_agent.observe (:360-373) and close_enough_to_goal (:425-432), one thread per navigator
__global__ __launch_bounds__(64) void mpc_observe_kernel(MpcArgs a, const float *__restrict__ ns, int32_t *cur_idx,
                                                         int32_t *actions_done, int give_up, int final_steps,
                                                         uint8_t *at_goal) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= a.P) return;
    float x[SSC_MAX_STATE];
#pragma unroll
    for (int k = 0; k < SSC_MAX_STATE; ++k) x[k] = (k < a.d) ? ns[p * a.d + k] : 0.0f;
    int idx = cur_idx[p];
    int done_act = actions_done[p];
    const bool goal = nav_observe_one(a, p, x, idx, done_act, give_up, final_steps);
    cur_idx[p] = idx;
    actions_done[p] = done_act;
    if (at_goal != nullptr) at_goal[p] = goal ? 1 : 0;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ssc

using namespace ssc;

extern "C" {

int ssc_mpc_sample_actions(int32_t P, int32_t N, int32_t H, int32_t act, const float *low, const float *high,
                           uint64_t seed, uint64_t problem_id0, uint64_t t, const uint64_t *d_t_base, float *d_A,
                           ssc_stream_t stream) {
    SSC_REQUIRE(P >= 0 && N >= 0 && H >= 0, "ssc_mpc_sample_actions: negative size");
    SSC_REQUIRE(act >= 1 && act <= SSC_MAX_ACT, "ssc_mpc_sample_actions: act_dim %d out of range", act);
    SSC_REQUIRE(low && high, "ssc_mpc_sample_actions: NULL bounds");
    if ((int64_t)P * N * H == 0) return SSC_OK;
    SSC_REQUIRE(d_A != nullptr, "ssc_mpc_sample_actions: d_A NULL");
    ActBounds bd{};
    for (int a = 0; a < act; ++a) {
        bd.low[a] = low[a];
        bd.span[a] = high[a] - low[a];
    }
    hipLaunchKernelGGL(mpc_sample_kernel, dim3(blocks_for((int64_t)P * N)), dim3(256), 0, as_stream(stream), P, N,
                       H, act, bd, seed, problem_id0, t, d_t_base, d_A);
    return check_launch("ssc_mpc_sample_actions");
}

size_t ssc_mpc_score_workspace_bytes(int32_t P, int32_t N, int32_t H) {
    if (P <= 0 || N <= 0 || H < 0) return 256;
    const size_t nblk = (size_t)(N + kMpcBlock - 1) / kMpcBlock;
    return align256((size_t)P * nblk * (H + 1) * 2 * sizeof(double)) + align256((size_t)P * nblk * 4) +
           align256((size_t)P * nblk * 4) + align256((size_t)P * 4);
}

static int mpc_score_common(const char *who, const ssc_mpc_problems *pr, const float *d_S, float *d_scores, int32_t *d_best_idx,
                            float *d_best_score, const SelectArgs &sel, void *d_workspace, size_t workspace_bytes,
                            ssc_stream_t stream) {
    SSC_REQUIRE(pr != nullptr, "%s: problems NULL", who);
    SSC_REQUIRE(pr->n_problems >= 0 && pr->n_samples >= 0, "%s: negative size", who);
    SSC_REQUIRE(pr->horizon >= 0 && pr->horizon + 1 <= kMaxH1, "%s: horizon %d > %d", who, pr->horizon,
                kMaxH1 - 1);
    SSC_REQUIRE(pr->state_dim >= 1 && pr->state_dim <= SSC_MAX_STATE, "%s: state_dim out of range", who);
    if (pr->n_problems == 0 || pr->n_samples == 0) return SSC_OK;
    SSC_REQUIRE(pr->wp && pr->left && pr->wp_off && pr->cur_idx && pr->radii && d_S && d_scores && d_best_idx,
                "%s: NULL device pointer", who);
    const size_t need = ssc_mpc_score_workspace_bytes(pr->n_problems, pr->n_samples, pr->horizon);
    SSC_REQUIRE(d_workspace && workspace_bytes >= need, "%s: workspace %zu < %zu", who, workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    MpcArgs a;
    a.P = pr->n_problems; a.N = pr->n_samples; a.H = pr->horizon; a.d = pr->state_dim;
    a.wp = pr->wp; a.left = pr->left; a.radii = pr->radii; a.wp_off = pr->wp_off; a.cur_idx = pr->cur_idx; a.plan_of = pr->plan_of; a.wp_len = pr->wp_len; a.active = pr->active;
    a.live_list = (pr->live_list != nullptr && pr->n_live != nullptr) ? pr->live_list : nullptr; a.n_live = pr->n_live;
    a.theta = pr->theta; a.gamma = pr->gamma; a.hpf = pr->horizontal_penalty_factor;
    a.per_row = pr->per_row_projection;
    a.nblk = (a.N + kMpcBlock - 1) / kMpcBlock;
    if (a.N <= 64) {
        // one launch: every problem lives inside one wave (mpc_small_kernel)
        const int G = a.N <= 16 ? 16 : a.N <= 32 ? 32 : 64;
        const int64_t blocks = ((int64_t)a.P + kMpcBlock / G - 1) / (kMpcBlock / G);
        SSC_REQUIRE(blocks <= 0x7fffffff, "%s: too many problems", who);
        const dim3 grid((unsigned)blocks);
#define SSC_SMALL16(D) mpc_small_kernel<16, D>
#define SSC_SMALL32(D) mpc_small_kernel<32, D>
#define SSC_SMALL64(D) mpc_small_kernel<64, D>
        if (G == 16) SSC_MPC_LAUNCH_D(a.d, SSC_SMALL16, grid, dim3(kMpcBlock), 0, s, a, sel, d_S, d_scores, d_best_idx, d_best_score);
        else if (G == 32) SSC_MPC_LAUNCH_D(a.d, SSC_SMALL32, grid, dim3(kMpcBlock), 0, s, a, sel, d_S, d_scores, d_best_idx, d_best_score);
        else SSC_MPC_LAUNCH_D(a.d, SSC_SMALL64, grid, dim3(kMpcBlock), 0, s, a, sel, d_S, d_scores, d_best_idx, d_best_score);
#undef SSC_SMALL16
#undef SSC_SMALL32
#undef SSC_SMALL64
        return check_launch(who);
    }
    SSC_REQUIRE((int64_t)a.P * a.nblk <= 0x7fffffff, "%s: too many blocks", who);
    char *w = static_cast<char *>(d_workspace);
    double *partial = reinterpret_cast<double *>(w);
    w += align256((size_t)a.P * a.nblk * (a.H + 1) * 2 * sizeof(double));
    float *bbs = reinterpret_cast<float *>(w);
    w += align256((size_t)a.P * a.nblk * 4);
    int32_t *bbi = reinterpret_cast<int32_t *>(w);
    w += align256((size_t)a.P * a.nblk * 4);
    int32_t *ticket = reinterpret_cast<int32_t *>(w);
    const size_t part_bytes = (size_t)a.nblk * (a.H + 1) * 2 * sizeof(double);
    const int stage_partials = part_bytes <= 32 * 1024;
    const size_t lds_b = stage_partials ? part_bytes : 0;
    const dim3 grid((unsigned)((int64_t)a.P * a.nblk));     // problem-major in x: P may exceed the gridDim.y limit
#define SSC_PASS_A(D) mpc_pass_a_kernel<D>
#define SSC_PASS_B(D) mpc_pass_b_kernel<D>
    SSC_MPC_LAUNCH_D(a.d, SSC_PASS_A, grid, dim3(kMpcBlock), 0, s, a, d_S, partial, ticket);
    SSC_MPC_LAUNCH_D(a.d, SSC_PASS_B, grid, dim3(kMpcBlock), lds_b, s, a, stage_partials, sel, d_S, partial, d_scores, bbs, bbi,
                     ticket, d_best_idx, d_best_score);
#undef SSC_PASS_A
#undef SSC_PASS_B
    return check_launch(who);
}

int ssc_mpc_score(const ssc_mpc_problems *pr, const float *d_S, float *d_scores, int32_t *d_best_idx,
                  float *d_best_score, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    SelectArgs sel{};
    return mpc_score_common("ssc_mpc_score", pr, d_S, d_scores, d_best_idx, d_best_score, sel, d_workspace, workspace_bytes,
                            stream);
}

int ssc_mpc_score_select(const ssc_mpc_problems *pr, const float *d_S, float *d_scores, int32_t *d_best_idx,
                         float *d_best_score, const float *d_A, const ssc_mpc_sampling *sp, int32_t act_dim,
                         float noise_amount, uint64_t noise_seed, uint64_t problem_id0, uint64_t t, float *d_action,
                         float *d_best_path, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    SSC_REQUIRE(pr != nullptr, "ssc_mpc_score_select: problems NULL");
    SSC_REQUIRE(act_dim >= 1 && act_dim <= SSC_MAX_ACT, "ssc_mpc_score_select: act_dim %d out of range", act_dim);
    SSC_REQUIRE((d_A != nullptr) != (sp != nullptr), "ssc_mpc_score_select: exactly one of d_A and sampling must be given");
    SSC_REQUIRE(pr->horizon >= 1, "ssc_mpc_score_select: horizon < 1");
    if (pr->n_problems == 0 || pr->n_samples == 0) return SSC_OK;
    SSC_REQUIRE(d_action != nullptr, "ssc_mpc_score_select: d_action NULL");
    SelectArgs sel{};
    sel.action = d_action; sel.best_path = d_best_path; sel.A = d_A; sel.act = act_dim; sel.noise = noise_amount;
    sel.seed = noise_seed; sel.pid0 = problem_id0; sel.t = t;
    if (sp != nullptr) {
        SSC_REQUIRE(sp->n_samples == pr->n_samples, "ssc_mpc_score_select: sampling.n_samples %d != problems.n_samples %d",
                    sp->n_samples, pr->n_samples);
        sel.s_seed = sp->seed; sel.s_pid0 = sp->problem_id0; sel.s_t = sp->t; sel.t_base = sp->d_t_base;
        for (int a = 0; a < act_dim; ++a) {
            sel.low[a] = sp->low[a];
            sel.span[a] = sp->high[a] - sp->low[a];
        }
    }
    return mpc_score_common("ssc_mpc_score_select", pr, d_S, d_scores, d_best_idx, d_best_score, sel, d_workspace,
                            workspace_bytes, stream);
}


int ssc_mpc_observe(const ssc_mpc_problems *pr, const float *d_new_state, int32_t *d_cur_idx,
                    int32_t *d_actions_done, int32_t give_up_after, int32_t final_steps, uint8_t *d_at_goal,
                    ssc_stream_t stream) {
    SSC_REQUIRE(pr != nullptr, "ssc_mpc_observe: problems NULL");
    SSC_REQUIRE(pr->n_problems >= 0, "ssc_mpc_observe: negative size");
    SSC_REQUIRE(pr->state_dim >= 1 && pr->state_dim <= SSC_MAX_STATE, "ssc_mpc_observe: state_dim out of range");
    if (pr->n_problems == 0) return SSC_OK;
    SSC_REQUIRE(pr->wp && pr->wp_off && pr->radii && d_new_state && d_cur_idx && d_actions_done,
                "ssc_mpc_observe: NULL device pointer");
    MpcArgs a{};
    a.P = pr->n_problems; a.d = pr->state_dim;
    a.wp = pr->wp; a.left = pr->left; a.radii = pr->radii; a.wp_off = pr->wp_off; a.cur_idx = pr->cur_idx; a.plan_of = pr->plan_of; a.wp_len = pr->wp_len; a.active = pr->active;
    a.live_list = (pr->live_list != nullptr && pr->n_live != nullptr) ? pr->live_list : nullptr; a.n_live = pr->n_live;
    a.theta = pr->theta;
    hipLaunchKernelGGL(mpc_observe_kernel, dim3((a.P + 63) / 64), dim3(64), 0, as_stream(stream), a, d_new_state,
                       d_cur_idx, d_actions_done, give_up_after, final_steps, d_at_goal);
    return check_launch("ssc_mpc_observe");
}

int ssc_mpc_select_action(int32_t P, int32_t N, int32_t H, int32_t d, int32_t act, const float *d_A,
                          const float *d_S, const int32_t *d_best_idx, float noise_amount, uint64_t seed,
                          uint64_t problem_id0, uint64_t t, float *d_action, float *d_best_path,
                          ssc_stream_t stream) {
    SSC_REQUIRE(P >= 0 && N >= 1 && H >= 1, "ssc_mpc_select_action: bad sizes");
    SSC_REQUIRE(d >= 1 && d <= SSC_MAX_STATE && act >= 1 && act <= SSC_MAX_ACT, "ssc_mpc_select_action: bad dims");
    if (P == 0) return SSC_OK;
    SSC_REQUIRE(d_A && d_S && d_best_idx && d_action, "ssc_mpc_select_action: NULL device pointer");
    hipLaunchKernelGGL(mpc_select_kernel, dim3((P + 63) / 64), dim3(64), 0, as_stream(stream), P, N, H, d, act, d_A,
                       d_S, d_best_idx, noise_amount, seed, problem_id0, t, d_action, d_best_path);
    return check_launch("ssc_mpc_select_action");
}

int ssc_nav_compact(int64_t n, const uint8_t *d_mode, int32_t *d_list, int32_t *d_count, ssc_stream_t stream) {
    SSC_REQUIRE(n >= 0 && n <= 0x7fffffffLL, "ssc_nav_compact: n = %lld", (long long)n);
    if (n == 0) return SSC_OK;
    SSC_REQUIRE(d_mode && d_list && d_count, "ssc_nav_compact: NULL device pointer");
    hipLaunchKernelGGL(nav_compact_kernel, dim3(blocks_for(n, kCompactBlock)), dim3(kCompactBlock), 0, as_stream(stream), n, d_mode, d_list, d_count);
    return check_launch("ssc_nav_compact");
}

}  // extern "C"
