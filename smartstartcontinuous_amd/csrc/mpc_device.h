// mpc_device.h -- pieces of the navigator shared by the scorer (mpc.hip) and the fused MPC rollout step
// (rollout.hip): the problem-set arguments, the radii-scaled distance and the waypoint bookkeeping of one env.
#pragma once
#include "ssc_device.h"

namespace ssc {

struct MpcArgs {
    int32_t P, N, H, d;
    const float *wp, *left, *radii;
    const int32_t *wp_off, *cur_idx;
    const int32_t *plan_of, *wp_len;   // optional plan pool (ssc_mpc_problems): problem p follows plan plan_of[p]
    const uint8_t *active;             // optional: problems whose byte is 0 are not scored (one-launch scorer)
    const int32_t *live_list, *n_live; // optional compact work list (one-launch scorer): groups serve live_list[0 .. *n_live)
    float theta, gamma, hpf;
    int32_t per_row;
    int32_t nblk;  // blocks per problem
};

// the plan problem p follows, and that plan's waypoint count
__device__ __forceinline__ int plan_index(const MpcArgs &a, int p) { return a.plan_of != nullptr ? a.plan_of[p] : p; }
__device__ __forceinline__ int plan_len(const MpcArgs &a, int q) {
    return a.plan_of != nullptr ? a.wp_len[q] : a.wp_off[q + 1] - a.wp_off[q];
}

// distance_func of numerical.py:116-124: || (x - y) / radii ||
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

// The plan a problem follows, as the bookkeeping below needs it: loaded on its own so that a caller can put these (dependent)
// loads in flight early -- they do not depend on the state the env is about to reach.
struct NavPlan {
    int off, W;
    float inv_r[SSC_MAX_STATE];
};
__device__ __forceinline__ NavPlan nav_plan_load(const MpcArgs &a, int p) {
    const int q = plan_index(a, p);
    NavPlan pl;
    pl.off = a.wp_off[q];
    pl.W = plan_len(a, q);
#pragma unroll
    for (int k = 0; k < SSC_MAX_STATE; ++k) pl.inv_r[k] = (k < a.d) ? 1.0f / a.radii[q * a.d + k] : 0.0f;
    return pl;
}

// NND_MB_agent.observe (:360-373) and close_enough_to_goal (:425-432) for a problem following plan `pl` at the new state x:
// advances idx / done_act in place, returns the at-goal flag.
__device__ __forceinline__ bool nav_observe_plan(const MpcArgs &a, const NavPlan &pl, const float *x, int &idx, int &done_act,
                                                 int give_up, int final_steps) {
    const int W = pl.W, d = a.d;
    const float *inv_r = pl.inv_r;
    const float *wp = a.wp + (int64_t)pl.off * d;
    idx = min(idx, max(W - 1, 0));   // a re-published (shorter) plan under a live env: see load_window (mpc.hip)
    // (the three waypoint rows are independent loads: all distances first, one round trip)
    const float dc = ell_dist(x, wp + idx * d, inv_r, d);                        // :364
    const float dn = ell_dist(x, wp + min(idx + 1, W - 1) * d, inv_r, d);        // :365
    const bool near = ell_dist(x, wp + (W - 1) * d, inv_r, d) <= a.theta;        // :426
    const bool move = (dc <= a.theta || dn <= dc) && idx != W - 1;               // :491-496
    if (move || (done_act > give_up && idx != W - 1)) {                          // :368-373
        idx += 1;
        done_act = 0;
    }
    return near || (idx == W - 1 && final_steps <= done_act);                     // :429-431
}

__device__ __forceinline__ bool nav_observe_one(const MpcArgs &a, int p, const float *x, int &idx, int &done_act,
                                                int give_up, int final_steps) {
    return nav_observe_plan(a, nav_plan_load(a, p), x, idx, done_act, give_up, final_steps);
}

}  // namespace ssc
