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
//   pass B  per block: fixed-order reduction of the partials -> G1_t, G2_t; per sample: replay the
//           trajectory, add progress and penalty terms -> score; block argmax (lowest index wins)
//   pass C  per problem: argmax over blocks.
// HBM traffic is the [H+1][P*N][d] trajectory read twice (8 B/sample-step for d=2) -- negligible
// next to the forward simulation that produced it.
#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

constexpr int kMpcBlock = 256;
constexpr int kMaxH1 = 33;  // horizon + 1 <= 33

struct MpcArgs {
    int32_t P, N, H, d;
    const float *wp, *left, *radii;
    const int32_t *wp_off, *cur_idx;
    float theta, gamma, hpf;
    int32_t per_row;
    int32_t nblk;  // blocks per problem
};

__device__ __forceinline__ float ell_dist(const float *x, const float *y, const float *inv_r, int d) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < SSC_MAX_STATE; ++k)
        if (k < d) {
            const float v = (x[k] - y[k]) * inv_r[k];
            s = fmaf(v, v, s);
        }
    return sqrtf(s);
}

// One sample's walk over the horizon.  PASS 0: accumulate a.b / b.b per t into sums[t][2].
// PASS 1: compute the score using the global G1/G2 per t.
template <int PASS>
__device__ __forceinline__ float mpc_walk(const MpcArgs &a, const float *__restrict__ S, int64_t row, int64_t M,
                                          const float *wps, const float *lefts, int W, int idx0,
                                          const float *inv_r, double (*sums)[2], const double (*G)[2]) {
    const int d = a.d;
    float pt[SSC_MAX_STATE];
    int idx = idx0;
    float score = 0.0f;
    float prev = 0.0f;
    float gpow = 1.0f;
    for (int t = 0; t <= a.H; ++t) {
        const float *p = S + ((int64_t)t * M + row) * d;
#pragma unroll
        for (int k = 0; k < SSC_MAX_STATE; ++k) pt[k] = (k < d) ? p[k] : 0.0f;
        if (t == 0) prev = lefts[idx] + ell_dist(pt, wps + idx * d, inv_r, d);  // NND_MB_agent.py:573-576
        const int nxt = min(idx + 1, W - 1);
        float dc = ell_dist(wps + idx * d, pt, inv_r, d);   // :589
        const float dn = ell_dist(wps + nxt * d, pt, inv_r, d);  // :590
        const bool move = (dc <= a.theta || dn <= dc) && idx != W - 1;  // move_to_next :491-496
        idx += move ? 1 : 0;                                // :600
        dc = move ? dn : dc;                                // :603
        const float end = lefts[idx] + dc;                  // :606
        if (PASS == 1) score += (prev - end) * gpow;        // :609
        prev = end;                                         // :610
        const int b = max(idx - 1, 0);                      // :615
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
        if (PASS == 0) {
            sums[t][0] += (double)ab;
            sums[t][1] += (double)bb;
        } else {
            // proj = (sum(a.b) / sum(b.b)) * b ; distance(proj, a)   (numerical.py:84-98 + :116-124)
            const float c = a.per_row ? ab / bb : (float)(G[t][0] / G[t][1]);
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < SSC_MAX_STATE; ++k)
                if (k < d) {
                    const float v = fmaf(c, bv[k], -av[k]);
                    s = fmaf(v, v, s);
                }
            score -= sqrtf(s) * a.hpf * a.gamma;  // :622 (gamma, not gamma^t)
            gpow *= a.gamma;
        }
    }
    return score;
}

__device__ __forceinline__ void load_problem(const MpcArgs &a, int p, float *wps, float *lefts, float *inv_r,
                                             int &W, int &idx0) {
    const int off = a.wp_off[p];
    W = a.wp_off[p + 1] - off;
    idx0 = a.cur_idx[p];
    for (int e = threadIdx.x; e < W * a.d; e += blockDim.x) wps[e] = a.wp[(int64_t)off * a.d + e];
    for (int e = threadIdx.x; e < W; e += blockDim.x) lefts[e] = a.left[off + e];
    if (threadIdx.x < a.d) inv_r[threadIdx.x] = 1.0f / a.radii[p * a.d + threadIdx.x];
    __syncthreads();
}

// dynamic LDS: wps [Wmax*d] | lefts [Wmax] | inv_r [8] | (pass B) G [(H+1)][2] doubles | reduction scratch
__global__ __launch_bounds__(kMpcBlock) void mpc_pass_a_kernel(MpcArgs a, int Wmax, const float *__restrict__ S,
                                                               double *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);  // [4 waves][H+1][2]
    float *wps = reinterpret_cast<float *>(red + 4 * kMaxH1 * 2);
    float *lefts = wps + Wmax * a.d;
    float *inv_r = lefts + Wmax;
    const int p = blockIdx.y;
    int W, idx0;
    load_problem(a, p, wps, lefts, inv_r, W, idx0);
    const int n = blockIdx.x * kMpcBlock + threadIdx.x;
    double sums[kMaxH1][2];
    for (int t = 0; t <= a.H; ++t) sums[t][0] = sums[t][1] = 0.0;
    if (n < a.N)
        mpc_walk<0>(a, S, (int64_t)p * a.N + n, (int64_t)a.P * a.N, wps, lefts, W, idx0, inv_r, sums, nullptr);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = 0; t <= a.H; ++t)
        for (int q = 0; q < 2; ++q) {
            double v = sums[t][q];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            if (lane == 0) red[(wave * kMaxH1 + t) * 2 + q] = v;
        }
    __syncthreads();
    for (int e = threadIdx.x; e < (a.H + 1) * 2; e += kMpcBlock) {
        const int t = e >> 1, q = e & 1;
        double v = 0.0;
        for (int w = 0; w < kMpcBlock / 64; ++w) v += red[(w * kMaxH1 + t) * 2 + q];
        partial[(((int64_t)p * a.nblk + blockIdx.x) * (a.H + 1) + t) * 2 + q] = v;
    }
}

__global__ __launch_bounds__(kMpcBlock) void mpc_pass_b_kernel(MpcArgs a, int Wmax, const float *__restrict__ S,
                                                               const double *__restrict__ partial,
                                                               float *__restrict__ scores,
                                                               float *__restrict__ blk_best_score,
                                                               int32_t *__restrict__ blk_best_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double(*G)[2] = reinterpret_cast<double(*)[2]>(smem);  // [H+1][2]
    float *rs = reinterpret_cast<float *>(G + kMaxH1);      // [4] block argmax scratch
    int *ri = reinterpret_cast<int *>(rs + 4);
    float *wps = reinterpret_cast<float *>(ri + 4);
    float *lefts = wps + Wmax * a.d;
    float *inv_r = lefts + Wmax;
    const int p = blockIdx.y;
    // fixed-order reduction of the per-block partials of this problem
    for (int e = threadIdx.x; e < (a.H + 1) * 2; e += kMpcBlock) {
        const int t = e >> 1, q = e & 1;
        double v = 0.0;
        for (int b = 0; b < a.nblk; ++b) v += partial[(((int64_t)p * a.nblk + b) * (a.H + 1) + t) * 2 + q];
        G[t][q] = v;
    }
    int W, idx0;
    load_problem(a, p, wps, lefts, inv_r, W, idx0);  // ends with __syncthreads()
    const int n = blockIdx.x * kMpcBlock + threadIdx.x;
    float score = -INFINITY;
    int best = 0x7fffffff;
    if (n < a.N) {
        score = mpc_walk<1>(a, S, (int64_t)p * a.N + n, (int64_t)a.P * a.N, wps, lefts, W, idx0, inv_r, nullptr, G);
        scores[(int64_t)p * a.N + n] = score;
        best = n;
        if (isnan(score)) score = -INFINITY;  // np.argmax would return the first NaN; we skip NaNs
    }
    // argmax, lowest index on ties (np.argmax, NND_MB_agent.py:626)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float os = __shfl_xor(score, m);
        const int oi = __shfl_xor(best, m);
        if (os > score || (os == score && oi < best)) { score = os; best = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { rs[wave] = score; ri[wave] = best; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kMpcBlock / 64; ++w)
            if (rs[w] > score || (rs[w] == score && ri[w] < best)) { score = rs[w]; best = ri[w]; }
        blk_best_score[p * a.nblk + blockIdx.x] = score;
        blk_best_idx[p * a.nblk + blockIdx.x] = best;
    }
}

__global__ __launch_bounds__(64) void mpc_pass_c_kernel(int P, int nblk, const float *__restrict__ blk_best_score,
                                                        const int32_t *__restrict__ blk_best_idx,
                                                        int32_t *__restrict__ best_idx,
                                                        float *__restrict__ best_score) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    float s = -INFINITY;
    int bi = 0x7fffffff;
    for (int b = 0; b < nblk; ++b) {
        const float os = blk_best_score[p * nblk + b];
        const int oi = blk_best_idx[p * nblk + b];
        if (os > s || (os == s && oi < bi)) { s = os; bi = oi; }
    }
    best_idx[p] = (bi == 0x7fffffff) ? 0 : bi;
    if (best_score) best_score[p] = s;
}

struct ActBounds {
    float low[SSC_MAX_ACT], span[SSC_MAX_ACT];
};

// npr.uniform(low, high, (N, H, act)) (NND_MB_agent.py:500-501); oracle: mpc_action_samples
__global__ __launch_bounds__(256) void mpc_sample_kernel(int P, int N, int H, int act, ActBounds bd, uint64_t seed,
                                                         uint64_t problem_id0, uint64_t t, float *__restrict__ A) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= (int64_t)P * N) return;
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
    const int off = a.wp_off[p], W = a.wp_off[p + 1] - off, d = a.d;
    float inv_r[SSC_MAX_STATE], x[SSC_MAX_STATE];
#pragma unroll
    for (int k = 0; k < SSC_MAX_STATE; ++k) {
        inv_r[k] = (k < d) ? 1.0f / a.radii[p * d + k] : 0.0f;
        x[k] = (k < d) ? ns[p * d + k] : 0.0f;
    }
    int idx = cur_idx[p];
    int done_act = actions_done[p];
    const float *wp = a.wp + (int64_t)off * d;
    const float dc = ell_dist(x, wp + idx * d, inv_r, d);                        // :364
    const float dn = ell_dist(x, wp + min(idx + 1, W - 1) * d, inv_r, d);        // :365
    const bool move = (dc <= a.theta || dn <= dc) && idx != W - 1;               // :491-496
    if (move || (done_act > give_up && idx != W - 1)) {                          // :368-373
        idx += 1;
        done_act = 0;
    }
    cur_idx[p] = idx;
    actions_done[p] = done_act;
    if (at_goal != nullptr) {
        const bool near = ell_dist(x, wp + (W - 1) * d, inv_r, d) <= a.theta;   // :426
        at_goal[p] = (near || (idx == W - 1 && final_steps <= done_act)) ? 1 : 0; // :429-431
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ssc

using namespace ssc;

extern "C" {

int ssc_mpc_sample_actions(int32_t P, int32_t N, int32_t H, int32_t act, const float *low, const float *high,
                           uint64_t seed, uint64_t problem_id0, uint64_t t, float *d_A, ssc_stream_t stream) {
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
                       H, act, bd, seed, problem_id0, t, d_A);
    return check_launch("ssc_mpc_sample_actions");
}

size_t ssc_mpc_score_workspace_bytes(int32_t P, int32_t N, int32_t H) {
    if (P <= 0 || N <= 0 || H < 0) return 256;
    const size_t nblk = (size_t)(N + kMpcBlock - 1) / kMpcBlock;
    return align256((size_t)P * nblk * (H + 1) * 2 * sizeof(double)) + align256((size_t)P * nblk * 4) +
           align256((size_t)P * nblk * 4);
}

int ssc_mpc_score(const ssc_mpc_problems *pr, const float *d_S, float *d_scores, int32_t *d_best_idx,
                  float *d_best_score, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream) {
    SSC_REQUIRE(pr != nullptr, "ssc_mpc_score: problems NULL");
    SSC_REQUIRE(pr->n_problems >= 0 && pr->n_samples >= 0, "ssc_mpc_score: negative size");
    SSC_REQUIRE(pr->horizon >= 0 && pr->horizon + 1 <= kMaxH1, "ssc_mpc_score: horizon %d > %d", pr->horizon,
                kMaxH1 - 1);
    SSC_REQUIRE(pr->state_dim >= 1 && pr->state_dim <= SSC_MAX_STATE, "ssc_mpc_score: state_dim out of range");
    if (pr->n_problems == 0 || pr->n_samples == 0) return SSC_OK;
    SSC_REQUIRE(pr->wp && pr->left && pr->wp_off && pr->cur_idx && pr->radii && d_S && d_scores && d_best_idx,
                "ssc_mpc_score: NULL device pointer");
    const size_t need = ssc_mpc_score_workspace_bytes(pr->n_problems, pr->n_samples, pr->horizon);
    SSC_REQUIRE(d_workspace && workspace_bytes >= need, "ssc_mpc_score: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t s = as_stream(stream);
    // the largest waypoint count decides the LDS carve: read wp_off back?  No host sync is allowed,
    // so the caller passes packed arrays and we size LDS for the maximum the kernel supports.
    const int Wmax = (int)((48 * 1024) / ((pr->state_dim + 1) * sizeof(float)));  // 48 KB of waypoints+left
    MpcArgs a;
    a.P = pr->n_problems; a.N = pr->n_samples; a.H = pr->horizon; a.d = pr->state_dim;
    a.wp = pr->wp; a.left = pr->left; a.radii = pr->radii; a.wp_off = pr->wp_off; a.cur_idx = pr->cur_idx;
    a.theta = pr->theta; a.gamma = pr->gamma; a.hpf = pr->horizontal_penalty_factor;
    a.per_row = pr->per_row_projection;
    a.nblk = (a.N + kMpcBlock - 1) / kMpcBlock;
    char *w = static_cast<char *>(d_workspace);
    double *partial = reinterpret_cast<double *>(w);
    w += align256((size_t)a.P * a.nblk * (a.H + 1) * 2 * sizeof(double));
    float *bbs = reinterpret_cast<float *>(w);
    w += align256((size_t)a.P * a.nblk * 4);
    int32_t *bbi = reinterpret_cast<int32_t *>(w);
    const size_t lds_common = (size_t)Wmax * (a.d + 1) * 4 + 8 * 4;
    const size_t lds_a = 4 * kMaxH1 * 2 * sizeof(double) + lds_common;
    const size_t lds_b = kMaxH1 * 2 * sizeof(double) + 8 * 4 + lds_common;
    const dim3 grid(a.nblk, a.P);
    hipLaunchKernelGGL(mpc_pass_a_kernel, grid, dim3(kMpcBlock), lds_a, s, a, Wmax, d_S, partial);
    hipLaunchKernelGGL(mpc_pass_b_kernel, grid, dim3(kMpcBlock), lds_b, s, a, Wmax, d_S, partial, d_scores, bbs, bbi);
    hipLaunchKernelGGL(mpc_pass_c_kernel, dim3((a.P + 63) / 64), dim3(64), 0, s, a.P, a.nblk, bbs, bbi, d_best_idx,
                       d_best_score);
    return check_launch("ssc_mpc_score");
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
    a.wp = pr->wp; a.left = pr->left; a.radii = pr->radii; a.wp_off = pr->wp_off; a.cur_idx = pr->cur_idx;
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

}  // extern "C"
