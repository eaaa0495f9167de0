// env_step.hip -- single-step / reset / observe kernels (the gym.Env-shaped API).
//
// These are the one-launch-per-step entry points behind VecEnv.step()/reset(); at
// 65 536 envs one step touches 1.6 MB and is launch/L2-bound -- the
// HBM-bound path is the fused K-step kernel in rollout.hip.  Elementwise, one thread per
// env, coalesced dword accesses over the SoA state columns.
#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

// Continuous_MountainCarEnv_Editted.step (continuous_mountain_car_editted.py:60-82) +
// gym TimeLimit.step (:154-159) when steps != nullptr.
__global__ __launch_bounds__(kBlock) void mc_step_kernel(McConst c, int64_t n, float *__restrict__ pos,
                                                         float *__restrict__ vel,
                                                         const float *__restrict__ act,
                                                         float *__restrict__ rew,
                                                         uint8_t *__restrict__ done,
                                                         int32_t *__restrict__ steps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float p = pos[i], v = vel[i], r;
    bool goal;
    mc_step_one(c, p, v, act[i], r, goal);
    bool d = goal;
    if (steps != nullptr) {
        const int32_t el = steps[i] + 1;
        steps[i] = el;
        d = d || (c.max_episode_steps > 0 && el >= c.max_episode_steps);
    }
    pos[i] = p;
    vel[i] = v;
    rew[i] = r;
    done[i] = d ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void pend_step_kernel(PendConst c, int64_t n, float *__restrict__ th,
                                                           float *__restrict__ thdot,
                                                           const float *__restrict__ act,
                                                           float *__restrict__ obs, float *__restrict__ rew,
                                                           uint8_t *__restrict__ done,
                                                           int32_t *__restrict__ steps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float t = th[i], td = thdot[i], r;
    pend_step_one(c, t, td, act[i], r);
    bool d = false;  // Pendulum never terminates by itself
    if (steps != nullptr) {
        const int32_t el = steps[i] + 1;
        steps[i] = el;
        d = c.max_episode_steps > 0 && el >= c.max_episode_steps;
    }
    th[i] = t;
    thdot[i] = td;
    rew[i] = r;
    done[i] = d ? 1 : 0;
    if (obs != nullptr) pend_observe_one(t, td, obs[i], obs[n + i], obs[2 * n + i]);
}

__global__ __launch_bounds__(kBlock) void env_reset_kernel(int kind, McConst mc, int64_t n,
                                                           const uint8_t *__restrict__ mask,
                                                           float *__restrict__ s0, float *__restrict__ s1,
                                                           int32_t *__restrict__ steps,
                                                           float *__restrict__ ep_ret, float *__restrict__ ou_x,
                                                           uint64_t seed, uint64_t env_id0, uint64_t t) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (mask != nullptr && mask[i] == 0) return;
    const u32x4 w = rng_words(seed, env_id0 + (uint64_t)i, t, TAG_RESET);
    float a, b;
    if (kind == SSC_ENV_MOUNTAINCAR)
        mc_reset_one(mc, w, a, b);
    else
        pend_reset_one(w, a, b);
    s0[i] = a;
    s1[i] = b;
    if (steps != nullptr) steps[i] = 0;
    if (ep_ret != nullptr) ep_ret[i] = 0.0f;
    if (ou_x != nullptr) ou_x[i] = 0.0f;
}

__global__ __launch_bounds__(kBlock) void env_observe_kernel(int kind, int64_t n, const float *__restrict__ s0,
                                                             const float *__restrict__ s1,
                                                             float *__restrict__ obs) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (kind == SSC_ENV_MOUNTAINCAR) {
        obs[i] = s0[i];
        obs[n + i] = s1[i];
    } else {
        pend_observe_one(s0[i], s1[i], obs[i], obs[n + i], obs[2 * n + i]);
    }
}

// One launch instead of 2*obs_dim+3 strided copies: thread e of column c copies VEC consecutive elements of the
// tail (VEC = 4: 16-byte accesses, n % 4 == 0 keeps a vector inside one step's row; the u8 column moves 4 bytes).
template <int VEC>
__global__ __launch_bounds__(kBlock) void pack_tail_kernel(ssc_transition_log log, int obs_dim, int K, int g,
                                                           int64_t n, const double *__restrict__ stats,
                                                           unsigned char *__restrict__ out) {
    const int64_t per = (int64_t)g * n;
    const int ncol = 2 * obs_dim + 2;             // fp32 columns; the u8 done column follows
    const int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * VEC;
    const int col = blockIdx.y;
    const int64_t step = e / n, i = e - step * n;
    const int64_t rs = log.row_stride ? log.row_stride : n, drs = log.done_row_stride ? log.done_row_stride : n;
    if (e < per) {
        if (col < ncol) {
            const int64_t src = (int64_t)(K - g + step) * rs + i;
            const float *p = col < obs_dim ? log.obs[col]
                             : col == obs_dim ? log.act
                             : col == obs_dim + 1 ? log.rew : log.obs2[col - obs_dim - 2];
            float *dst = reinterpret_cast<float *>(out) + (int64_t)col * per + e;
            if (VEC == 4) *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(p + src);
            else *dst = p[src];
        } else {
            const uint8_t *ps = log.done + (int64_t)(K - g + step) * drs + i;
            unsigned char *dst = out + (int64_t)ncol * per * 4 + e;
            if (VEC == 4) *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(ps);
            else *dst = *ps;
        }
    }
    if (stats != nullptr && col == 0 && blockIdx.x == 0 && threadIdx.x < 4) {
        const int64_t off = (((int64_t)ncol * per * 4 + per) + 7) & ~(int64_t)7;
        reinterpret_cast<double *>(out + off)[threadIdx.x] = stats[threadIdx.x];
    }
}

// |3*pos| must stay inside cos_bounded's domain.
int validate_mc_params(const ssc_env_params *p, const char *who) {
    if (p->kind != SSC_ENV_MOUNTAINCAR) return set_error(SSC_EINVAL, "%s: params are not MountainCar", who);
    if (!(p->min_position < p->max_position) || p->min_position < -1.5f || p->max_position > 1.5f)
        return set_error(SSC_EUNSUPPORTED, "%s: position range [%g, %g] outside [-1.5, 1.5]", who,
                         (double)p->min_position, (double)p->max_position);
    if (!(p->max_speed > 0.0f) || !(p->min_action <= p->max_action))
        return set_error(SSC_EINVAL, "%s: bad max_speed / action range", who);
    return SSC_OK;
}


// reduce_epsilon / reduce_eta (DDPG_Baselines_agent.py:77-78, smartexplorationcontinuous.py:372-376) once per finished
// GENERATION of episodes, where the episode counter lives.  One thread; the arithmetic is the host's Python float (fp64).
struct DecayArgs {
    const double *finished;
    double finished0, per_generation;
    int32_t n_sched;
    double factor[4], floor_[4];
    double *state;
    float *out[4];
};

__global__ void decay_schedule_kernel(DecayArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double target = floor((*a.finished - a.finished0) / a.per_generation);
    const double applied = a.state[0];
    int64_t todo = (int64_t)(target - applied);
    if (todo < 0) todo = 0;
    for (int i = 0; i < a.n_sched; ++i) {
        double v = a.state[1 + i];
        for (int64_t g = 0; g < todo; ++g) {
            const double nv = fmax(v * a.factor[i], a.floor_[i]);
            if (nv == v) break;  // reached its floor (or factor 1): further generations change nothing
            v = nv;
        }
        a.state[1 + i] = v;
        if (a.out[i] != nullptr) *a.out[i] = (float)v;
    }
    if (todo > 0) a.state[0] = applied + (double)todo;
}

}  // namespace ssc

using namespace ssc;

extern "C" {

int ssc_mc_step(const ssc_env_params *p, int64_t n, float *d_pos, float *d_vel, const float *d_act,
                float *d_rew, uint8_t *d_done, int32_t *d_steps, ssc_stream_t stream) {
    SSC_REQUIRE(p != nullptr, "ssc_mc_step: params NULL");
    SSC_REQUIRE(n >= 0, "ssc_mc_step: n = %lld < 0", (long long)n);
    if (int rc = validate_mc_params(p, "ssc_mc_step")) return rc;
    if (n == 0) return SSC_OK;
    SSC_REQUIRE(d_pos && d_vel && d_act && d_rew && d_done, "ssc_mc_step: NULL device pointer");
    hipLaunchKernelGGL(mc_step_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream),
                       make_mc_const(*p), n, d_pos, d_vel, d_act, d_rew, d_done, d_steps);
    return check_launch("ssc_mc_step");
}

int ssc_pend_step(const ssc_env_params *p, int64_t n, float *d_th, float *d_thdot, const float *d_act,
                  float *d_obs, float *d_rew, uint8_t *d_done, int32_t *d_steps, ssc_stream_t stream) {
    SSC_REQUIRE(p != nullptr, "ssc_pend_step: params NULL");
    SSC_REQUIRE(p->kind == SSC_ENV_PENDULUM, "ssc_pend_step: params are not Pendulum");
    SSC_REQUIRE(n >= 0, "ssc_pend_step: n = %lld < 0", (long long)n);
    if (n == 0) return SSC_OK;
    SSC_REQUIRE(d_th && d_thdot && d_act && d_rew && d_done, "ssc_pend_step: NULL device pointer");
    hipLaunchKernelGGL(pend_step_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream),
                       make_pend_const(*p), n, d_th, d_thdot, d_act, d_obs, d_rew, d_done, d_steps);
    return check_launch("ssc_pend_step");
}

int ssc_env_reset(const ssc_env_params *p, int64_t n, const uint8_t *d_mask, float *d_s0, float *d_s1,
                  int32_t *d_steps, float *d_ep_ret, float *d_ou_x, uint64_t seed, uint64_t env_id0,
                  uint64_t t, ssc_stream_t stream) {
    SSC_REQUIRE(p != nullptr, "ssc_env_reset: params NULL");
    SSC_REQUIRE(p->kind == SSC_ENV_MOUNTAINCAR || p->kind == SSC_ENV_PENDULUM, "ssc_env_reset: bad kind");
    SSC_REQUIRE(n >= 0, "ssc_env_reset: n = %lld < 0", (long long)n);
    if (n == 0) return SSC_OK;
    SSC_REQUIRE(d_s0 && d_s1, "ssc_env_reset: NULL state pointer");
    hipLaunchKernelGGL(env_reset_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream), p->kind,
                       make_mc_const(*p), n, d_mask, d_s0, d_s1, d_steps, d_ep_ret, d_ou_x, seed, env_id0, t);
    return check_launch("ssc_env_reset");
}

size_t ssc_pack_bytes(int32_t obs_dim, int32_t g, int64_t n) {
    if (obs_dim < 1 || obs_dim > SSC_MAX_OBS || g < 0 || n < 0) return 0;
    const size_t per = (size_t)g * (size_t)n;
    return ((per * (8 * (size_t)obs_dim + 9) + 7) & ~(size_t)7) + 4 * sizeof(double);
}

int ssc_pack_transitions(const ssc_transition_log *log, int32_t obs_dim, int32_t K, int32_t g, int64_t n,
                         const double *d_stats, void *d_out, ssc_stream_t stream) {
    SSC_REQUIRE(log != nullptr && d_out != nullptr, "ssc_pack_transitions: NULL argument");
    SSC_REQUIRE(obs_dim >= 1 && obs_dim <= SSC_MAX_OBS, "ssc_pack_transitions: obs_dim %d", obs_dim);
    SSC_REQUIRE(g >= 0 && g <= K && n >= 0, "ssc_pack_transitions: need 0 <= g <= K, n >= 0");
    if (g == 0 || n == 0) return SSC_OK;
    for (int c = 0; c < obs_dim; ++c) SSC_REQUIRE(log->obs[c] && log->obs2[c], "ssc_pack_transitions: NULL column");
    SSC_REQUIRE(log->act && log->rew && log->done, "ssc_pack_transitions: NULL column");
    // 16-byte path: rows and every column base 16-byte aligned (the u8 column: 4-byte)
    const int64_t rs = log->row_stride ? log->row_stride : n, drs = log->done_row_stride ? log->done_row_stride : n;
    bool vec = (n % 4 == 0) && (rs % 4 == 0) && (drs % 4 == 0) && (reinterpret_cast<uintptr_t>(d_out) % 16 == 0) &&
               (reinterpret_cast<uintptr_t>(log->act) % 16 == 0) && (reinterpret_cast<uintptr_t>(log->rew) % 16 == 0) &&
               (reinterpret_cast<uintptr_t>(log->done) % 4 == 0);
    for (int c = 0; c < obs_dim; ++c)
        vec = vec && (reinterpret_cast<uintptr_t>(log->obs[c]) % 16 == 0) && (reinterpret_cast<uintptr_t>(log->obs2[c]) % 16 == 0);
    if (vec) {
        const dim3 grid(blocks_for((int64_t)g * n / 4), 2 * obs_dim + 3);
        hipLaunchKernelGGL(pack_tail_kernel<4>, grid, dim3(kBlock), 0, as_stream(stream), *log, obs_dim, K, g, n, d_stats,
                           static_cast<unsigned char *>(d_out));
    } else {
        const dim3 grid(blocks_for((int64_t)g * n), 2 * obs_dim + 3);
        hipLaunchKernelGGL(pack_tail_kernel<1>, grid, dim3(kBlock), 0, as_stream(stream), *log, obs_dim, K, g, n, d_stats,
                           static_cast<unsigned char *>(d_out));
    }
    return check_launch("ssc_pack_transitions");
}

int ssc_env_observe(const ssc_env_params *p, int64_t n, const float *d_s0, const float *d_s1,
                    float *d_obs, ssc_stream_t stream) {
    SSC_REQUIRE(p != nullptr, "ssc_env_observe: params NULL");
    SSC_REQUIRE(p->kind == SSC_ENV_MOUNTAINCAR || p->kind == SSC_ENV_PENDULUM, "ssc_env_observe: bad kind");
    SSC_REQUIRE(n >= 0, "ssc_env_observe: n < 0");
    if (n == 0) return SSC_OK;
    SSC_REQUIRE(d_s0 && d_s1 && d_obs, "ssc_env_observe: NULL device pointer");
    hipLaunchKernelGGL(env_observe_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream), p->kind, n,
                       d_s0, d_s1, d_obs);
    return check_launch("ssc_env_observe");
}


int ssc_decay_schedule(const double *d_finished, double finished0, double per_generation, int32_t n_sched, const double *factor,
                       const double *floor, double *d_state, float *const *d_out, ssc_stream_t stream) {
    SSC_REQUIRE(d_finished != nullptr && d_state != nullptr, "ssc_decay_schedule: NULL device pointer");
    SSC_REQUIRE(n_sched >= 0 && n_sched <= 4, "ssc_decay_schedule: n_sched = %d not in 0..4", n_sched);
    SSC_REQUIRE(per_generation > 0.0, "ssc_decay_schedule: per_generation must be positive");
    if (n_sched == 0) return SSC_OK;
    SSC_REQUIRE(factor != nullptr && floor != nullptr, "ssc_decay_schedule: NULL factor / floor array");
    DecayArgs a{};
    a.finished = d_finished; a.finished0 = finished0; a.per_generation = per_generation; a.n_sched = n_sched; a.state = d_state;
    for (int i = 0; i < n_sched; ++i) {
        SSC_REQUIRE(factor[i] >= 0.0, "ssc_decay_schedule: negative factor");
        a.factor[i] = factor[i]; a.floor_[i] = floor[i];
        a.out[i] = d_out != nullptr ? d_out[i] : nullptr;
    }
    hipLaunchKernelGGL(decay_schedule_kernel, dim3(1), dim3(64), 0, as_stream(stream), a);
    return check_launch("ssc_decay_schedule");
}

}  // extern "C"
