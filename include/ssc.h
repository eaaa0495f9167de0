/* ssc.h -- C ABI of libssc.so, the MI355X (gfx950) rollout engine behind the
 * gym.Env / RLAgent surface of darren-huang/SmartStartContinuous.
 *
 * The reference is pure Python (no FFI of its own); each entry point below names the
 * reference interface it replaces (file:line relative to the reference root).  The
 * Python host layer (smartstartcontinuous_amd/_ffi.py) binds exactly these symbols with
 * ctypes; INTEGRATION.md shows the stubs a reference maintainer would add.
 *
 * Conventions
 *   - Every pointer named d_* / inside the descriptor structs marked "device" is a DEVICE
 *     pointer owned by the caller (a PyTorch-ROCm tensor's data_ptr()).  The library
 *     allocates nothing on the device and keeps no state between calls; all RNG is
 *     counter-based (Philox4x32-10 keyed by seed / global env id / global step).
 *   - All work is enqueued on the given hipStream_t (passed as void*); no call
 *     synchronises, none uses the default stream implicitly.
 *   - Return 0 (SSC_OK) or a negative SSC_E* code; ssc_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI, nothing exits.
 *   - Arrays are structure-of-arrays, fp32 unless stated; "[K][n]" means K rows of n
 *     contiguous elements.
 */
#ifndef SSC_H
#define SSC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSC_VERSION 108 /* 0.1.8: ssc_ddpg_desc.critic_l2_reg / clip_norm, ssc_mse_batches; 0.1.7: LayerNorm (ssc_actor_desc / ssc_critic_desc ln*, ssc_ddpg_desc.layer_norm); 0.1.6: ssc_nav_compact + live lists in ssc_mpc_sampling / ssc_mpc_problems; 0.1.5: ssc_ou_desc.d_epsilon, ssc_decay_schedule, ssc_replay_append_shard; 0.1.4: ssc_path_shortcut; 0.1.3: ssc_zscore_concat (0.1.2: plan pool + active mask in ssc_mpc_problems / ssc_mpc_sampling, ssc_smartstart_rollout_step) */

typedef void *ssc_stream_t; /* hipStream_t */

enum {
    SSC_OK = 0,
    SSC_EINVAL = -1,       /* bad argument (null pointer, negative size, ...) */
    SSC_EUNSUPPORTED = -2, /* shape / option outside what the kernels implement */
    SSC_EHIP = -3          /* a HIP runtime call failed; see ssc_last_error() */
};

enum { SSC_ENV_MOUNTAINCAR = 0, SSC_ENV_PENDULUM = 1 };
enum { SSC_POLICY_RANDOM = 0, SSC_POLICY_ACTOR = 1 };
enum { SSC_PREC_F32 = 0, SSC_PREC_BF16_MFMA = 1, SSC_PREC_BF16_MFMA_PREPARED = 2 };

#define SSC_MAX_OBS 3
#define SSC_MAX_LAYERS 4
#define SSC_MAX_STATE 8
#define SSC_MAX_ACT 4

/* Environment constants.
 * MountainCar: Continuous_MountainCarEnv_Editted.__init__
 *   (smartstart/environments/continuous_mountain_car_editted.py:35-54) + the gym TimeLimit
 *   of make_timed_env (:154-159).
 * Pendulum: gym 0.10.5 PendulumEnv.__init__ [third-party, pinned in Pipfile.lock]. */
typedef struct ssc_env_params {
    int32_t kind;              /* SSC_ENV_* */
    int32_t max_episode_steps; /* TimeLimit; <= 0 means no limit */
    /* MountainCar */
    float min_action, max_action;
    float min_position, max_position;
    float max_speed, goal_position, power;
    float reset_low, reset_high; /* reset(): pos ~ U(reset_low, reset_high), vel = 0 (:84-86) */
    /* Pendulum */
    float max_torque, pend_max_speed, dt, g, m, l;
    int32_t pend_v1_order; /* 0: gym 0.10.5 "v0" update order, 1: modern "v1" order */
} ssc_env_params;

/* DDPG actor (Actor_Editted.__call__, DDPG_Baselines_editted/models_editted.py:38-61).
 * Weights are fp32 device arrays in TensorFlow layout W[in][out]. */
typedef struct ssc_actor_desc {
    int32_t obs_dim, h1, h2, act_dim;
    const float *W1, *b1, *W2, *b2, *W3, *b3; /* device */
    int32_t last_layer_tanh;                  /* models_editted.py:53-56 */
    int32_t precision;                        /* SSC_PREC_*: hidden GEMM in fp32 VALU or bf16 MFMA */
    float obs_clip;                           /* > 0: the observation is clipped to [-obs_clip, obs_clip] before layer 1 --
                                                 DDPG_editted feeds its networks tf.clip_by_value(obs, observation_range)
                                                 (ddpg_editted.py:106-109, observation_range = (-5, 5) in every run);
                                                 0: no clip (bare Actor_Editted.__call__) */
    /* layer_norm=True (models_editted.py:45-46, 50-51; the class default, unused by the shipped runs):
     * tc.layers.layer_norm(center=True, scale=True) [third-party TF 1.5: over the units of a row, variance epsilon 1e-12]
     * behind each hidden dense layer, in front of its activation.  gamma / beta of layer 1 [h1] and layer 2 [h2], device;
     * all four NULL: no LayerNorm.  LayerNorm networks run on the fp32 kernels (`precision` must be SSC_PREC_F32). */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
} ssc_actor_desc;

/* Exploration noise of DDPG_editted.pi (ddpg_editted.py:266-271):
 * DecayingOrnsteinUhlenbeckActionNoise (smartstart/RLAgents/DDPG_Baselines_agent.py:52-78). */
typedef struct ssc_ou_desc {
    float mu, sigma, theta, dt;
    float epsilon; /* current epsilon (decayed per episode by the host, :77-78); 0 disables noise */
    const float *d_epsilon; /* not NULL: the current epsilon is this DEVICE float, read when the kernel starts (`epsilon` is
                               ignored and the noise path always runs) -- so that a loop whose decay is computed on the
                               device (ssc_decay_schedule) never needs the host between two rollouts */
} ssc_ou_desc;

typedef struct ssc_policy_desc {
    int32_t kind;             /* SSC_POLICY_* */
    float act_low, act_high;  /* action_space bounds (Policy_Random, policy_random.py:9-15;
                                 DDPG_Baselines_agent.scale, DDPG_Baselines_agent.py:236-240) */
    ssc_actor_desc actor;     /* SSC_POLICY_ACTOR */
    ssc_ou_desc ou;           /* SSC_POLICY_ACTOR */
} ssc_policy_desc;

/* Per-env persistent state, each [n] (device).  s0/s1 = (position, velocity) for
 * MountainCar, (theta, theta_dot) for Pendulum. */
typedef struct ssc_rollout_state {
    float *s0, *s1;
    int32_t *steps; /* TimeLimit's elapsed-step counter of the running episode */
    float *ep_ret;  /* running episode return (Episode.total_reward, datacontainers.py:41-49) */
    float *ou_x;    /* OU-noise state x_prev; may be NULL for SSC_POLICY_RANDOM */
} ssc_rollout_state;

/* Transition log of one rollout chunk: the record ReplayBuffer.add stores
 * (smartstart/RLAgents/replay_buffer.py:49-74, tuple (s, a, r, t, s2) :53), as SoA
 * columns of [K][n].  obs/obs2 hold obs_dim columns (2 for MountainCar, 3 for Pendulum);
 * unused entries may be NULL. */
typedef struct ssc_transition_log {
    float *obs[SSC_MAX_OBS];
    float *act;
    float *rew;
    uint8_t *done;
    float *obs2[SSC_MAX_OBS];
    int64_t row_stride;      /* elements between step k and k+1 of one fp32 column; 0 means n (dense [K][n]).
                                A packed chunk puts all fp32 columns of a step side by side:
                                row_stride = (2*obs_dim+2)*n, column c starts at base + c*n. */
    int64_t done_row_stride; /* same for the u8 done column; 0 means n */
} ssc_transition_log;

/* Completed-episode records: what Summary.append keeps per episode
 * (smartstart/utilities/datacontainers.py:173-193: (len(episode), total_reward)). */
typedef struct ssc_episode_ring {
    int64_t *env_id;  /* [capacity] global env id */
    int32_t *length;  /* [capacity] */
    float *ret;       /* [capacity] */
    uint32_t *cursor; /* [1] device counter; records beyond capacity are dropped but counted */
    int32_t capacity;
} ssc_episode_ring;

int ssc_version(void);
const char *ssc_last_error(void);

/* Fill *p with the reference defaults.  kind = SSC_ENV_MOUNTAINCAR: power = 0.0015 *
 * power_scalar (continuous_mountain_car_editted.py:43). */
int ssc_env_params_default(int kind, float power_scalar, int32_t max_episode_steps, ssc_env_params *p);

/* One env.step for n envs -- Continuous_MountainCarEnv_Editted.step
 * (continuous_mountain_car_editted.py:60-82) wrapped by gym TimeLimit.step when d_steps is
 * given (d_steps[i] += 1; done |= d_steps[i] >= max_episode_steps).  pos/vel are updated
 * in place.  d_steps may be NULL (bare env, no time limit). */
int ssc_mc_step(const ssc_env_params *p, int64_t n, float *d_pos, float *d_vel, const float *d_act,
                float *d_rew, uint8_t *d_done, int32_t *d_steps, ssc_stream_t stream);

/* gym 0.10.5 PendulumEnv.step for n envs [third-party]; d_obs is [3][n] =
 * (cos th, sin th, thdot) of the NEW state, may be NULL. */
int ssc_pend_step(const ssc_env_params *p, int64_t n, float *d_th, float *d_thdot, const float *d_act,
                  float *d_obs, float *d_rew, uint8_t *d_done, int32_t *d_steps, ssc_stream_t stream);

/* env.reset for the envs selected by d_mask (NULL = all) --
 * continuous_mountain_car_editted.py:84-86 / gym PendulumEnv.reset.  Draws come from
 * Philox(seed; env_id0 + i, t, TAG_RESET); also zeroes d_steps / d_ep_ret / d_ou_x entries
 * (each may be NULL). */
int ssc_env_reset(const ssc_env_params *p, int64_t n, const uint8_t *d_mask, float *d_s0, float *d_s1,
                  int32_t *d_steps, float *d_ep_ret, float *d_ou_x, uint64_t seed, uint64_t env_id0,
                  uint64_t t, ssc_stream_t stream);

/* Observation of the current state: MountainCar [2][n] = (pos, vel); Pendulum [3][n]. */
int ssc_env_observe(const ssc_env_params *p, int64_t n, const float *d_s0, const float *d_s1,
                    float *d_obs, ssc_stream_t stream);

/* K fused steps of the rlTrain inner loop (smartstart/reinforcementLearningCore/rlTrain.py:75-100)
 * for n independent envs: get_action -> env.step -> record -> auto-reset on done.
 * One thread owns one env; state stays in registers for the K steps.
 *   policy RANDOM = Policy_Random.get_action (NN_Dynamics_Model/policy_random.py:14-15), i.e. the
 *     reference's collect_samples_threaded.py:52-111 rollout;
 *   policy ACTOR  = DDPG_Baselines_agent.get_action (DDPG_Baselines_agent.py:206-234).
 * log, ring, d_stats may each be NULL.  d_stats is double[4] on the device and is
 * ACCUMULATED into: {sum of rewards, goal terminations, env-steps, finished episodes}.
 * Global step index of the first step is step0; env i has global id env_id0 + i, so any
 * sharding of the id space reproduces the same per-env streams. */
int ssc_rollout(const ssc_env_params *p, const ssc_policy_desc *policy, int64_t n, int32_t K,
                const ssc_rollout_state *state, const ssc_transition_log *log,
                const ssc_episode_ring *ring, double *d_stats, uint64_t seed, uint64_t env_id0,
                uint64_t step0, ssc_stream_t stream);

/* The per-episode decay of the exploration schedules, computed where the episode counter lives: the reference decays
 * epsilon (DecayingOrnsteinUhlenbeckActionNoise.reduce_epsilon, DDPG_Baselines_agent.py:77-78, called from end_episode
 * :255-258) and eta (SmartStartContinuous.reduce_eta, smartexplorationcontinuous.py:372-376) once per finished episode; a
 * loop over n parallel envs decays once per GENERATION = `per_generation` finished episodes.  One thread:
 *   g = floor((*d_finished - finished0) / per_generation);  for every schedule i < n_sched, (g - applied) times:
 *   value_i = max(value_i * factor[i], floor[i])   in fp64, exactly the host's Python arithmetic;
 * d_state = double[1 + n_sched]: {generations applied, value_0, ...} (caller-initialised: 0, start values);
 * d_out[i] (device float, may be NULL) receives (float)value_i -- e.g. the ssc_ou_desc.d_epsilon of the next rollout.
 * d_finished: a device double, e.g. d_stats + 3 of ssc_rollout (finished episodes), finished0 its value when the loop
 * started.  n_sched <= 4; factor / floor are HOST arrays. */
int ssc_decay_schedule(const double *d_finished, double finished0, double per_generation, int32_t n_sched, const double *factor,
                       const double *floor, double *d_state, float *const *d_out, ssc_stream_t stream);

/* Packs the LAST g steps of a transition log into one contiguous buffer for the per-chunk exchange
 * (the rollout gather of NN_Dynamics_Model/collect_samples_threaded.py:31-50 as ONE RCCL message):
 *   d_out = [obs (obs_dim x g x n) f32 | act (g x n) | rew (g x n) | obs2 (obs_dim x g x n) | done (g x n) u8]
 * followed, 8-byte aligned, by a snapshot of d_stats (4 doubles) when d_stats != NULL.
 * d_out must hold ssc_pack_bytes(obs_dim, g, n) bytes. */
size_t ssc_pack_bytes(int32_t obs_dim, int32_t g, int64_t n);
int ssc_pack_transitions(const ssc_transition_log *log, int32_t obs_dim, int32_t K, int32_t g, int64_t n,
                         const double *d_stats, void *d_out, ssc_stream_t stream);

/* Batched actor forward, act[m][act_dim] = Actor_Editted(obs[m][obs_dim])
 * (models_editted.py:38-61); row-major in/out.  No noise, no clipping. */
int ssc_actor_forward(const ssc_actor_desc *actor, int64_t m, const float *d_obs, float *d_act,
                      ssc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * SmartStart navigator: NND_MB dynamics model + MPC (smartstart/RLAgents/NND_MB_agent.py)
 * ------------------------------------------------------------------------------------- */

/* feedforward_network (NN_Dynamics_Model/feedforward_network.py:3-23):
 * h = relu(h @ W_i + b_i) for the num_fc_layers hidden layers, z = h @ W_out + b_out.
 * n_layers = num_fc_layers + 1 weight matrices; dims = {in, depth, ..., out};
 * W[l] is fp32 [dims[l]][dims[l+1]] (TensorFlow layout), device. */
typedef struct ssc_mlp_desc {
    int32_t n_layers;
    int32_t dims[SSC_MAX_LAYERS + 1];
    const float *W[SSC_MAX_LAYERS];
    const float *b[SSC_MAX_LAYERS];
} ssc_mlp_desc;

/* z-score statistics of NND_MB_agent.__init__ (NND_MB_agent.py:302-315), by value. */
typedef struct ssc_norm {
    float mean_x[SSC_MAX_STATE], std_x[SSC_MAX_STATE];
    float mean_y[SSC_MAX_ACT], std_y[SSC_MAX_ACT];
    float mean_z[SSC_MAX_STATE], std_z[SSC_MAX_STATE];
} ssc_norm;

/* y[m][out] = feedforward_network(x[m][in]).  precision SSC_PREC_F32: every layer in fp32
 * (VALU); SSC_PREC_BF16_MFMA: hidden-layer contractions on bf16 MFMA with fp32 accumulation
 * (num_fc_layers 1 or 2, depth <= 512).  d_workspace: ssc_mlp_workspace_bytes() bytes. */
size_t ssc_mlp_workspace_bytes(const ssc_mlp_desc *mlp, int64_t m, int precision);
int ssc_mlp_forward(const ssc_mlp_desc *mlp, int64_t m, const float *d_x, float *d_y, int precision,
                    void *d_workspace, size_t workspace_bytes, ssc_stream_t stream);

/* Dyn_Model.do_forward_sim, batched branch (NN_Dynamics_Model/dynamics_model.py:204-240):
 *   S[0] = s0;  for t < H:  x = nan_to_num((S[t]-mean_x)/std_x) || nan_to_num((A[:,t]-mean_y)/std_y)
 *                           S[t+1] = S[t] + net(x) * std_z + mean_z
 * d_s0 is [s0_rows][state_dim]: s0_rows == 1 (one start state tiled to all rows, :215-217), s0_rows == m,
 * or any divisor P of m (P problems with m/P candidate sequences each: row r starts from state r / (m/P)).
 * d_A is [m][H][act_dim]; d_S is [H+1][m][state_dim]. */
size_t ssc_dyn_workspace_bytes(const ssc_mlp_desc *mlp, int64_t m, int precision);
/* Dyn_Model.run_validation's loss (dynamics_model.py:173-196 with mse_ of :42): d_batch_loss[b] = mean over the
 * `batch_elems` (= batchsize * out_dim) consecutive elements of batch b of (d_z - d_pred)^2, d_mean[0] = their mean in
 * batch order.  d_pred: the network's outputs for the first n_batches * batchsize rows (ssc_mlp_forward). */
int ssc_mse_batches(const float *d_pred, const float *d_z, int64_t n_batches, int64_t batch_elems, float *d_batch_loss,
                    float *d_mean, ssc_stream_t stream);

/* The MFMA path works from a packed bf16 image of the weights (and of the statistics) in the workspace.
 * ssc_dyn_forward_sim / ssc_mlp_forward with SSC_PREC_BF16_MFMA write that image on every call; a caller
 * whose weights stay put between calls -- the navigator between two Dyn_Model.train() rounds
 * (NND_MB_agent.py:421-423), like TF variables that live in the session across sess.run calls
 * (dynamics_model.py:226-233) -- writes it ONCE with ssc_dyn_prepare() and then passes
 * SSC_PREC_BF16_MFMA_PREPARED with the same workspace (which nothing else may touch in between).
 * norm may be NULL for ssc_mlp_forward use.  Workspace size: ssc_dyn_workspace_bytes(.., SSC_PREC_BF16_MFMA). */
int ssc_dyn_prepare(const ssc_mlp_desc *mlp, const ssc_norm *norm, void *d_workspace, size_t workspace_bytes,
                    ssc_stream_t stream);
int ssc_dyn_forward_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, int64_t m, int32_t H,
                        int32_t state_dim, int32_t act_dim, const float *d_s0, int64_t s0_rows,
                        const float *d_A, float *d_S, int precision, void *d_workspace,
                        size_t workspace_bytes, ssc_stream_t stream);

/* A set of P independent MPC problems (one per real env) with N candidate action sequences
 * each; rows of every [P*N] array are problem-major.  Waypoints / distances_left come from
 * NND_MB_agent.start_new_episode_plan (NND_MB_agent.py:375-423), packed back to back:
 * problem p owns rows wp_off[p] .. wp_off[p+1]-1 (at least 2 waypoints each). */
typedef struct ssc_mpc_problems {
    int32_t n_problems, n_samples, horizon, state_dim;
    const float *wp;        /* device [wp_off[P]][state_dim]  desired_states */
    const float *left;      /* device [wp_off[P]]             distances_left (:411-418) */
    const int32_t *wp_off;  /* device [P+1] */
    const int32_t *cur_idx; /* device [P]  current_desired_state_index */
    const float *radii;     /* device [P][state_dim] (numerical.py:61-62) */
    float theta, gamma, horizontal_penalty_factor; /* :143, :62 */
    int32_t per_row_projection; /* 0 = reference behaviour (batch-global np.sum, numerical.py:89-92);
                                   1 = corrected per-sample projection (NOT the reference) */
    /* Optional plan POOL (the vectorised SmartStart loop: many envs follow one of a few stored plans).  Both NULL:
     * problem p owns plan p, as described above.  Otherwise problem p follows plan q = plan_of[p]: its waypoints are
     * rows wp_off[q] .. wp_off[q] + wp_len[q] - 1 of wp / left and its radii row q of radii; cur_idx stays per problem. */
    const int32_t *plan_of; /* device [P] or NULL */
    const int32_t *wp_len;  /* device [number of plans] or NULL (required with plan_of) */
    /* Optional: device [P] bytes; a problem whose byte is 0 is not scored by the one-launch scorer (n_samples <= 64): its
     * scores / best_idx entries are left untouched.  NULL: every problem is scored. */
    const uint8_t *active;
    /* Optional compact work list (ssc_nav_compact), as in ssc_mpc_sampling: the one-launch scorer serves the *n_live problems
     * live_list[0 .. *n_live) and nothing else; outputs stay indexed by problem. */
    const int32_t *live_list;
    const int32_t *n_live;
} ssc_mpc_problems;

/* all_samples = npr.uniform(low, high, (N, H, act)) (NND_MB_agent.py:500-501) for P problems:
 * d_A [P*N][H][act_dim], Philox(seed; (problem_id0+p) << 32 | n, t, TAG_MPC).  d_t_base (may be NULL) is a
 * device-resident step counter added to t, so that a captured HIP graph can be replayed step after step. */
int ssc_mpc_sample_actions(int32_t n_problems, int32_t n_samples, int32_t horizon, int32_t act_dim,
                           const float *low, const float *high, uint64_t seed, uint64_t problem_id0,
                           uint64_t t, const uint64_t *d_t_base, float *d_A, ssc_stream_t stream);

/* The candidate action sequences of an MPC step as a SPECIFICATION instead of a matrix: all_samples =
 * npr.uniform(low, high, (N, H, act)) (NND_MB_agent.py:500-501) is a pure function of (seed, problem, sample, t) --
 * the Philox stream of ssc_mpc_sample_actions -- so the consumers can draw the numbers where they use them and the
 * [P*N][H][act] matrix need not exist. */
typedef struct ssc_mpc_sampling {
    int32_t n_samples;                       /* N: row r of an [P*N] array is sample r % N of problem r / N */
    float low[SSC_MAX_ACT], high[SSC_MAX_ACT];
    uint64_t seed, problem_id0, t;
    const uint64_t *d_t_base;                /* device step counter added to t (HIP-graph replay); may be NULL */
    /* Optional: device [number of problems] bytes; the rows of a problem whose byte is 0 need not be simulated (their part of
     * d_S / d_A_out is then left untouched).  The kernels with LDS-resident weights (depth <= 128, or one hidden layer) skip
     * a wave whose rows all belong to such problems; the streamed-W2 kernel ignores the mask.  NULL: every problem is live. */
    const uint8_t *d_problem_active;
    /* Optional COMPACT work list (ssc_nav_compact): device int32 list of the live problems and device count.  With both given
     * the fused small-network kernel assigns thread block i * 256 + j to row (list[slot] * N + sample) of the slot-th LIVE
     * problem -- the launch still covers P * N threads (HIP-graph safe), blocks past *d_n_live exit at once -- and every
     * array keeps its problem-major layout.  Other kernels ignore the list (they simulate every row). */
    const int32_t *d_live_list;
    const int32_t *d_n_live;
} ssc_mpc_sampling;

/* ssc_mpc_sample_actions + ssc_dyn_forward_sim in ONE launch (Dyn_Model.do_forward_sim fed by get_best_sim_actions,
 * NND_MB_agent.py:498-512): the forward-simulation kernel draws every row's action sequence itself (bit-identical to
 * ssc_mpc_sample_actions with the same specification), so the sample launch and the [m][H][act] read disappear.
 * d_A_out [m][H][act] (may be NULL) receives the sequences for callers that index them later (ssc_mpc_rollout_step);
 * ssc_mpc_score_select regenerates the winner's first action from the specification and does not need it.
 * precision SSC_PREC_F32 runs the two-launch equivalent and needs d_A_out. */
int ssc_mpc_forward_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, const ssc_mpc_sampling *sampling, int64_t m,
                        int32_t H, int32_t state_dim, int32_t act_dim, const float *d_s0, int64_t s0_rows,
                        float *d_A_out, float *d_S, int precision, void *d_workspace, size_t workspace_bytes,
                        ssc_stream_t stream);

/* generate_scores_add_delta (NND_MB_agent.py:566-628) + argmax (:625-626) per problem.
 * d_S [H+1][P*N][state_dim] (output of ssc_dyn_forward_sim); d_scores [P*N]; d_best_idx [P]
 * (index within the problem, lowest index on ties like np.argmax); d_best_score [P]. */
size_t ssc_mpc_score_workspace_bytes(int32_t n_problems, int32_t n_samples, int32_t horizon);
int ssc_mpc_score(const ssc_mpc_problems *prob, const float *d_S, float *d_scores, int32_t *d_best_idx,
                  float *d_best_score, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream);

/* ssc_mpc_score + ssc_mpc_select_action in the same two launches: the block that finishes a problem's argmax also
 * writes action[p] = best_sequence[0] + noise_amount * N(0,1) (no clip, NND_MB_agent.py:353-356; the draws of
 * ssc_mpc_select_action) and the predicted path S[:, best] -> d_best_path [P][H+1][state_dim] (may be NULL).  The
 * first action of the winner comes from d_A [P*N][H][act] when given, otherwise it is regenerated from `sampling`
 * (exactly one of the two must be non-NULL).  noise_seed / problem_id0 / t (+ *d_t_base of `sampling` when given)
 * key the noise like ssc_mpc_select_action. */
int ssc_mpc_score_select(const ssc_mpc_problems *prob, const float *d_S, float *d_scores, int32_t *d_best_idx,
                         float *d_best_score, const float *d_A, const ssc_mpc_sampling *sampling, int32_t act_dim,
                         float noise_amount, uint64_t noise_seed, uint64_t problem_id0, uint64_t t, float *d_action,
                         float *d_best_path, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream);

/* NND_MB_agent.observe (NND_MB_agent.py:360-373) + close_enough_to_goal (:425-432) for P navigators at
 * once: given the state each env reached, advance its waypoint index when the waypoint was reached /
 * overtaken (move_to_next, :491-496) or given up on (d_actions_done > give_up_after), and flag envs that
 * are within theta of their final waypoint (or timed out on it after final_steps actions).
 * d_cur_idx [P] and d_actions_done [P] are updated in place; d_at_goal [P] (u8) may be NULL.
 * d_new_state is [P][state_dim].  (get_action's `actions_done += 1`, :340, is the caller's.) */
int ssc_mpc_observe(const ssc_mpc_problems *prob, const float *d_new_state, int32_t *d_cur_idx,
                    int32_t *d_actions_done, int32_t give_up_after, int32_t final_steps, uint8_t *d_at_goal,
                    ssc_stream_t stream);

/* get_action_with_predicted_states (NND_MB_agent.py:339-358): action[p] = A[best][0] +
 * noise_amount * N(0,1) (no clip, :353-356); also copies the predicted path
 * S[:, best] -> d_best_path [P][H+1][state_dim] (may be NULL). */
int ssc_mpc_select_action(int32_t n_problems, int32_t n_samples, int32_t horizon, int32_t state_dim,
                          int32_t act_dim, const float *d_A, const float *d_S, const int32_t *d_best_idx,
                          float noise_amount, uint64_t seed, uint64_t problem_id0, uint64_t t,
                          float *d_action, float *d_best_path, ssc_stream_t stream);

/* One MPC-policy step of rollout(K, policy = 'mpc') for P envs (one navigation problem each), everything
 * after the scoring in ONE launch: executed action = best_sequence[0] + noise_amount * N(0,1) (NND_MB_agent.py:
 * 353-356, same draws as ssc_mpc_select_action), env.step (+ TimeLimit), transition-log row, chunk statistics,
 * episode record, NND_MB_agent.observe on the new observation (waypoint advance, :360-373), auto-reset of finished
 * envs (state, episode counters, navigator plan position back to d_start_idx) and the planning state of the
 * next step, d_plan_state [P][obs_dim] (the observation after a possible reset).
 * The global step t and the log row k are DEVICE counters (*d_t, *d_k), advanced by the launch itself, so a
 * HIP graph {ssc_mpc_sample_actions(.., 0, d_t, ..), ssc_dyn_forward_sim(d_plan_state, s0_rows = P),
 * ssc_mpc_score, ssc_mpc_rollout_step} is replayed K times per chunk without touching the host.
 * d_ticket: one zero-initialised int32 (election of the block that advances the counters). */
typedef struct ssc_mpc_nav_state {
    int32_t *cur_idx;            /* [P] current waypoint (== ssc_mpc_problems.cur_idx) */
    const int32_t *start_idx;    /* [P] waypoint a fresh episode starts from */
    int32_t *actions_done;       /* [P] actions spent on the current waypoint */
    uint8_t *at_goal;            /* [P] close_enough_to_goal (:425-432); may be NULL */
    int32_t give_up_after, final_steps;
} ssc_mpc_nav_state;

int ssc_mpc_rollout_step(const ssc_env_params *p, const ssc_mpc_problems *problems, const ssc_mpc_nav_state *nav,
                         const float *d_A, const int32_t *d_best_idx, float noise_amount, uint64_t noise_seed,
                         uint64_t problem_id0, const ssc_rollout_state *state, const ssc_transition_log *log,
                         const ssc_episode_ring *ring, double *d_stats, uint64_t env_seed, uint64_t env_id0,
                         uint64_t *d_t, int32_t *d_k, int32_t *d_ticket, float *d_plan_state, ssc_stream_t stream);

/* One step of the VECTORISED SmartStartContinuous loop (smartstart/smartexploration/smartexplorationcontinuous.py:
 * 307-376) for P envs, everything after the scoring in ONE launch -- ssc_mpc_rollout_step with a per-env mode:
 *   mode[p] = 1  "smart_start_pathing": the executed action is the navigator's (best_sequence[0] + noise, as above), the
 *                waypoint bookkeeping runs on the new observation, and close_enough_to_goal hands the env over to
 *                the base agent (mode 0) for the rest of the episode (:333-339);
 *   mode[p] = 0  the base agent acts: d_actor_out[p] (Actor_Editted on the clipped planning state, ssc_actor_forward)
 *                + epsilon * OU noise, clip, scale twice -- DDPG_Baselines_agent.get_action, the OU state advancing
 *                only on these steps, stream and arithmetic of ssc_rollout's ACTOR policy.
 * A finished episode (goal / TimeLimit) resets the env and the OU state and runs start_new_episode (:341-370): with
 * probability *d_eta (uniform draw: word x of Philox(env_seed; env id, t, tag 8); plan: word y) the env takes a plan out of the pool
 * -- slot (pool[0] + word % pool[1]) % pool[2] of the plan arrays behind problems->plan_of, if pool[1] > 0 -- starts it
 * at waypoint 0 and navigates unless the reset state is already close enough to the plan's goal.
 * d_eta, d_ou_epsilon (one float each) and d_pool (int32[3]: first slot, plans on offer, slots) are DEVICE values, so a
 * captured HIP graph of the step keeps following the host's decay schedule and pool refreshes.  d_mode_log (may be
 * NULL) receives the mode each env ACTED in, row *d_k of a [K][P] byte matrix.  problems->plan_of must be writable. */
typedef struct ssc_smartstart_step {
    uint8_t *mode;               /* [P] */
    int32_t *plan_of;            /* [P] == problems->plan_of */
    const float *d_actor_out;    /* [P] */
    const float *d_eta, *d_ou_epsilon;
    const int32_t *d_pool;       /* [3] */
    ssc_ou_desc ou;              /* epsilon field unused (d_ou_epsilon) */
    float act_low, act_high;
    uint8_t *d_mode_log;         /* [K][P] or NULL */
    int64_t mode_log_stride;     /* row stride of d_mode_log (0: P) */
    int32_t *d_n_live;           /* optional: the counter of ssc_nav_compact, zeroed by this launch (the last one of a step) so
                                    that the next step's compaction starts from 0 without a launch of its own */
} ssc_smartstart_step;

/* The envs that are navigating (d_mode[i] != 0) as a compact list: d_list[0 .. *d_count) = their indices (any order),
 * *d_count their number.  *d_count must be 0 on entry (ssc_smartstart_step.d_n_live leaves it so).  One pass: ballot +
 * prefix inside a wave, one atomic per wave for its base. */
int ssc_nav_compact(int64_t n, const uint8_t *d_mode, int32_t *d_list, int32_t *d_count, ssc_stream_t stream);

int ssc_smartstart_rollout_step(const ssc_env_params *p, const ssc_mpc_problems *problems, const ssc_mpc_nav_state *nav,
                                const ssc_smartstart_step *ss, const float *d_A, const int32_t *d_best_idx,
                                float noise_amount, uint64_t noise_seed, uint64_t problem_id0,
                                const ssc_rollout_state *state, const ssc_transition_log *log, const ssc_episode_ring *ring,
                                double *d_stats, uint64_t env_seed, uint64_t env_id0, uint64_t *d_t, int32_t *d_k,
                                int32_t *d_ticket, float *d_plan_state, ssc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * SmartStart selection (smartstart/smartexploration/smartexplorationcontinuous.py:223-305)
 * ------------------------------------------------------------------------------------- */

/* DDPG critic (Critic_Editted.__call__, DDPG_Baselines_editted/models_editted.py:78-100):
 *   x = relu(obs @ W1 + b1); x = concat(x, action); x = tanh|relu(x @ W2 + b2); q = x @ W3 + b3
 * W2 is [h1 + act_dim][h2]; weights fp32, TensorFlow layout, device. */
typedef struct ssc_critic_desc {
    int32_t obs_dim, act_dim, h1, h2;
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int32_t last_layer_tanh;
    float obs_clip;             /* as in ssc_actor_desc (ddpg_editted.py:106-109); 0: no clip */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b; /* LayerNorm (models_editted.py:85-86, 91-92) as in ssc_actor_desc; NULL: none */
} ssc_critic_desc;

/* q[m] = Critic(obs[m][obs_dim], act[m][act_dim]) -- the batched get_q_value of
 * ddpg_editted.py:274-279 once act = Actor(obs) (ssc_actor_forward). */
int ssc_critic_forward(const ssc_critic_desc *critic, int64_t m, const float *d_obs, const float *d_act,
                       float *d_q, ssc_stream_t stream);

/* scipy.stats.gaussian_kde(dataset).evaluate(points) [third-party, call site
 * smartexplorationcontinuous.py:260,275] for a bandwidth matrix chosen by the caller:
 *   pdf[i] = norm * sum_j exp(-0.5 * || Wh (points[i] - data[j]) ||^2)
 * Wh is the d x d row-major whitening matrix (chol(inv(covariance))^T, host array), norm =
 * 1 / (n * sqrt(det(2 pi covariance))).  data [n][d], points [m][d], pdf [m]; d <= SSC_MAX_STATE. */
int ssc_kde_evaluate(int32_t d, int64_t n, const float *d_data, int64_t m, const float *d_points,
                     const float *whitening, double norm, float *d_pdf, ssc_stream_t stream);

/* UCB1 over the candidate smart-start states (smartexplorationcontinuous.py:275-280):
 *   ucb[i] = alpha * value[i] + sqrt(beta * ln(D) / (D * pdf[i] * volume));  best = argmax (lowest
 * index on ties).  d_ucb [m] may be NULL; d_best [1]. */
int ssc_ucb_argmax(int64_t m, const float *d_value, const float *d_pdf, float alpha, float beta, double buffer_len,
                   double volume, float *d_ucb, int32_t *d_best, ssc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * DDPG training step
 * ------------------------------------------------------------------------------------- */

/* DDPG_editted.train() + update_target_net() (DDPG_Baselines_editted/ddpg_editted.py:287-339) as
 * driven by DDPG_Baselines_agent.train (smartstart/RLAgents/DDPG_Baselines_agent.py:264-273), in the
 * configuration of every shipped run: no observation/return normalisation, no popart, no l2
 * regularisation, no gradient clipping.  Each network's parameters are ONE flat fp32 device array
 * in TensorFlow trainable_vars order [W1 | b1 | W2 | b2 | W3 | b3] (the order U.flatgrad and
 * MpiAdam use); Adam moments have the same shape.  Critic W2 is [critic_h1 + act_dim][critic_h2]. */
typedef struct ssc_ddpg_desc {
    int32_t obs_dim, act_dim, actor_h1, actor_h2, critic_h1, critic_h2;
    int32_t last_layer_tanh;
    int32_t batch_size;                                   /* 64 in every shipped run; 1..4096 through ssc_ddpg_train_ws */
    float *actor, *critic, *target_actor, *target_critic; /* device, flat */
    float *adam_m_actor, *adam_v_actor, *adam_m_critic, *adam_v_critic; /* device, flat, zero-initialised */
    int32_t *adam_t;                                      /* device [2]: MpiAdam step counters (actor, critic) */
    float gamma, tau, actor_lr, critic_lr;
    float beta1, beta2, epsilon;                          /* MpiAdam: 0.9, 0.999, 1e-8 (ddpg_editted.py:176,198) */
    float obs_clip;                                       /* > 0: obs0 / obs1 clipped to [-obs_clip, obs_clip] before every
                                                             network (ddpg_editted.py:106-109); 0: no clip */
    int32_t layer_norm;                                   /* 1: actor and critic carry LayerNorm (models_editted.py:45-46,50-51,
                                                             85-86,91-92): every flat vector is then
                                                             [W1|b1|beta1|gamma1|W2|b2|beta2|gamma2|W3|b3] (TF trainable_vars order;
                                                             tc.layers.layer_norm creates beta before gamma) and the step runs on
                                                             the multi-workgroup kernels (ssc_ddpg_train_ws) */
    float critic_l2_reg;                                  /* > 0: critic_loss += critic_l2_reg * sum(W^2) / 2 over the critic's
                                                             three dense kernels (ddpg_editted.py:183-191; the name filter keeps
                                                             all of them, models_editted.py names no layer 'output'); 0: none */
    float clip_norm;                                      /* > 0: every variable's gradient, actor and critic, clipped to this
                                                             2-norm (U.flatgrad(..., clip_norm), ddpg_editted.py:175, 197);
                                                             <= 0: none.  Either option runs the multi-workgroup kernels. */
} ssc_ddpg_desc;

/* Replay storage the batches are drawn from: row-major device arrays of `capacity` records
 * (the (s, a, r, t, s2) tuple of replay_buffer.py:53). */
typedef struct ssc_replay_view {
    const float *s, *a, *r;
    const uint8_t *t;
    const float *s2;
    int64_t capacity;
} ssc_replay_view;

/* The same storage, writable: a device-resident replay ring fed straight from rollout chunks.  Record
 * number j (counted since the ring was created) lives at row j % capacity -- the FIFO of the reference's
 * deque (replay_buffer.py:53-72): once full, the oldest record is overwritten.  act_dim must be 1. */
typedef struct ssc_replay_ring {
    float *s, *a, *r;
    uint8_t *t;
    float *s2;
    int64_t capacity;
    int32_t obs_dim, act_dim;
    /* Episode index (both NULL: not kept).  The reference keeps episode_starting_indices beside its deque
     * (replay_buffer.py:33-44, 109-115, 209-213) so that the path to any stored state can be recovered; a chunk of n
     * parallel envs interleaves n episodes, so the device ring keeps, per record, the number of steps of ITS env's
     * running episode up to and including the record (ep_steps [capacity], 1 = the episode's first recorded step).
     * Records of one env lie n record numbers apart (record number = steps * n + env), hence the episode of record j
     * is records j - (ep_steps-1)*n, ..., j - n, j -- valid for path recovery while the first of them is still in the
     * ring.  ep_run [n] carries every env's running step count from one append to the next (zero-initialised; the
     * caller zeroes it when it skips steps between two appends). */
    int32_t *ep_steps;
    int32_t *ep_run;
} ssc_replay_ring;

/* ReplayBuffer.add for a whole rollout chunk (replay_buffer.py:49-74; called per step from
 * DDPG_Baselines_agent.observe, DDPG_Baselines_agent.py:238-240, which scales the reward): appends the
 * K*n records of `log` ([K][n] SoA columns as ssc_rollout wrote them; record order = step-major, then env)
 * after the `start` records appended so far.  The caller keeps the running count (start += K*n); it is
 * deterministic, so no device counter and no host sync is needed. */
int ssc_replay_append(const ssc_replay_ring *ring, const ssc_transition_log *log, int32_t K, int64_t n,
                      int64_t start, float reward_scale, ssc_stream_t stream);

/* The same append for ONE SHARD of a step: the chunk holds envs [env_off, env_off + n) of n_total envs per step, its
 * record (k, e) gets record number start + k * n_total + env_off + e -- the learner of a sharded run appends every
 * rank's gathered records this way and ends up with the ring of the single-GPU run, whatever the world size.  `start`
 * counts whole steps of n_total records; K * n_total <= capacity.  ep_run (if kept) is indexed by env_off + e. */
int ssc_replay_append_shard(const ssc_replay_ring *ring, const ssc_transition_log *log, int32_t K, int64_t n,
                            int64_t start, int64_t n_total, int64_t env_off, float reward_scale, ssc_stream_t stream);

/* ReplayBuffer.sample_batch (replay_buffer.py:79-91: random.sample, i.e. uniform WITHOUT replacement
 * inside a batch) for n_batches batches at once: d_idx [n_batches][batch_size] row indices in
 * [0, size), size = min(records appended, capacity) >= batch_size, batch_size <= 4096.
 * Philox(seed; counter0 + batch, attempt << 8 | slot, TAG_REPLAY) for batch_size <= 64 (one wave per batch),
 * attempt << 16 | slot above (one workgroup per batch); oracle: replay_sample_indices. */
int ssc_replay_sample(uint64_t seed, uint64_t counter0, int64_t size, int32_t n_batches, int32_t batch_size,
                      int32_t *d_idx, ssc_stream_t stream);

/* ReplayBuffer.get_possible_smart_start_indices (replay_buffer.py:136-152: random.sample(range(first, len), n_ss), first =
 * the oldest episode start still in the buffer) on the device ring: up to n_ss DISTINCT buffer indices (0 = oldest
 * record, size-1 = newest; size = min(count, capacity)) drawn uniformly from the records whose whole episode prefix is
 * still in the ring.  Rounds of Philox(seed; counter, round << 20 | slot, TAG_SMART_START) candidates; in a round a slot
 * keeps its candidate unless it is invalid, already taken, or wanted by a lower slot (deterministic; oracle:
 * smart_start_indices).  d_idx [n_ss] int32 (unfilled slots -1), *d_n = number of indices delivered (< n_ss only when
 * fewer valid records exist or after max_rounds = 64).  n_ss <= 4096.  count = records appended so far, n = envs per
 * step of the appended chunks.  Workspace: ssc_replay_smart_start_workspace_bytes(n_ss). */
size_t ssc_replay_smart_start_workspace_bytes(int32_t n_ss);
int ssc_replay_smart_start_indices(const ssc_replay_ring *ring, int64_t count, int64_t n, int32_t n_ss, uint64_t seed,
                                   uint64_t counter, int32_t *d_idx, int32_t *d_n, void *d_workspace,
                                   size_t workspace_bytes, ssc_stream_t stream);

/* ReplayBuffer.get_episodic_path_to_buffer_index (replay_buffer.py:154-176): the states of the episode that contains
 * buffer index *d_buffer_index (device int32; e.g. an entry of d_idx above) up to that record, plus its s2:
 * d_path [max_len + 1][obs_dim], *d_len = number of rows written (episode steps so far + 1; 0 when the record's
 * episode start has left the ring; at most the newest max_len steps of the prefix are kept). */
int ssc_replay_episode_path(const ssc_replay_ring *ring, int64_t count, int64_t n, const int32_t *d_buffer_index,
                            int32_t max_len, float *d_path, int32_t *d_len, ssc_stream_t stream);

/* n_iters sequential training iterations in ONE launch (one workgroup: the iterations are a serial
 * chain through the parameters).  d_batch_idx [n_iters][batch_size] are record indices
 * (ReplayBuffer.sample_batch, replay_buffer.py:79-91, draws them on the host).  d_losses
 * [n_iters][2] = (critic_loss, actor_loss) per iteration, may be NULL.
 * Shapes: batch 64, act_dim 1.  The shipped 64-32 actor / critic with a 2- or 3-d observation runs a kernel compiled
 * for that shape; any other layer sizes <= 64 run a step interpreter as long as the batch's activations plus the four
 * parameter vectors fit the 160 KB of LDS (64-32 with obs_dim <= 8 does; 64-64 does not) -- otherwise SSC_EUNSUPPORTED
 * with the byte count in ssc_last_error(). */
int ssc_ddpg_train(const ssc_ddpg_desc *ddpg, const ssc_replay_view *replay, const int32_t *d_batch_idx,
                   int32_t n_iters, float *d_losses, ssc_stream_t stream);

/* The same training step for ANY layer sizes and batch sizes 1..4096 -- the reference's own grid of actor / critic
 * 64-32, 128-64, 200-100 (DDPG_Baselines_agent.py:86-92, data/ddpg_baselines_summaries/hidden_layer_size_experiment/)
 * and the large batches of a vectorised actor-learner loop.  Shapes the single-workgroup kernels cover (batch 64,
 * layers <= 64) still run there; everything else runs multi-workgroup: the batch is tiled over workgroups of 16 rows
 * (whole forward / backward chain per tile, weights streamed from L2), per-workgroup gradient partials are summed in
 * workgroup order by a second launch that also applies MpiAdam and the soft target update -- two launches per
 * iteration, bitwise reproducible.  d_workspace: ssc_ddpg_train_workspace_bytes(ddpg) bytes of device memory (the
 * gradient partials; caller-owned like every buffer).  SSC_EUNSUPPORTED only when a 16-row tile's activations exceed
 * the 160 KB of LDS (roughly: 2 h1 + 2 h2 of the actor + 2 h1 + 4 h2 of the critic > 2500 units). */
size_t ssc_ddpg_train_workspace_bytes(const ssc_ddpg_desc *ddpg);
int ssc_ddpg_train_ws(const ssc_ddpg_desc *ddpg, const ssc_replay_view *replay, const int32_t *d_batch_idx,
                      int32_t n_iters, float *d_losses, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Dynamics-model training step
 * ------------------------------------------------------------------------------------- */

/* One iteration of Dyn_Model.train's inner loop (NN_Dynamics_Model/dynamics_model.py:98-101 /
 * :111-113): mse = mean((z - net(x))^2) over the batch (:41), tf.train.AdamOptimizer(lr) step
 * (:44-50) [third-party TF 1.5: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * theta -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + 1e-8)].  The batch is rows d_idx[0..B) of the
 * (already normalised) device data sets d_X [n][in], d_Z [n][out] -- the old/new mixing of :60-96 is
 * index arithmetic done by the caller.  Weights and Adam moments are updated in place. */
typedef struct ssc_mlp_train_desc {
    int32_t n_layers;
    int32_t dims[SSC_MAX_LAYERS + 1];
    float *W[SSC_MAX_LAYERS], *b[SSC_MAX_LAYERS];      /* device, updated in place */
    float *mW[SSC_MAX_LAYERS], *vW[SSC_MAX_LAYERS];    /* device Adam moments, zero-initialised */
    float *mb[SSC_MAX_LAYERS], *vb[SSC_MAX_LAYERS];
    int32_t *adam_t;                                    /* device [1] step counter */
    float lr, beta1, beta2, epsilon;
} ssc_mlp_train_desc;

size_t ssc_mlp_train_workspace_bytes(const ssc_mlp_train_desc *net, int32_t batch);
/* n_steps consecutive iterations enqueued by one call: step k trains on rows d_idx[k*batch .. (k+1)*batch) and
 * writes its batch MSE (before the update) to d_loss[k] (d_loss may be NULL).  One hidden layer of <= 512 units
 * (in <= 12, out <= 8 -- the shapes the reference ships) runs ONE fused launch per step; other shapes run the
 * generic chain of per-layer launches. */
int ssc_mlp_train_steps(const ssc_mlp_train_desc *net, const float *d_X, const float *d_Z, const int32_t *d_idx,
                        int32_t batch, int32_t n_steps, float *d_loss, void *d_workspace, size_t workspace_bytes,
                        ssc_stream_t stream);
/* = ssc_mlp_train_steps with n_steps = 1. */
int ssc_mlp_train_step(const ssc_mlp_train_desc *net, const float *d_X, const float *d_Z, const int32_t *d_idx,
                       int32_t batch, float *d_loss, void *d_workspace, size_t workspace_bytes, ssc_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Random-rollout data collection -> dynamics-model training set (SURVEY.md section 8a row A17 and the
 * data format either side of it): CollectSamples.collect_samples / do_rollout
 * (NN_Dynamics_Model/collect_samples_threaded.py:25-111), generate_training_data_inputs / _outputs
 * (NN_Dynamics_Model/data_manipulation.py:58-88), the z-score statistics of NND_MB_agent.py:302-319 and
 * helper_funcs.add_noise (NN_Dynamics_Model/helper_funcs.py:10-17).
 * ------------------------------------------------------------------------------------- */

/* A "rollout" is one env's FIRST episode segment of the chunk `log` ([K][n] SoA columns written by
 * ssc_rollout from freshly reset envs): it stops after its first terminal step like do_rollout (:89-93).
 * d_len [n]  = number of steps of rollout i (first done inclusive, else K);
 * d_off [n+1] = exclusive prefix sum of max(d_len - 1, 0): the first data-set row of rollout i, d_off[n] the
 * total -- every rollout loses its last entry (data_manipulation.py:66-74). */
size_t ssc_dataset_scan_workspace_bytes(int64_t n);
int ssc_dataset_scan(const ssc_transition_log *log, int32_t K, int64_t n, int32_t *d_len, int64_t *d_off,
                     void *d_workspace, size_t workspace_bytes, ssc_stream_t stream);

/* Row-major data set, rollout after rollout like np.concatenate over the list of rollouts (:77-78, :87):
 * d_X [rows][obs_dim] = s_i, d_Y [rows][1] = a_i, d_Z [rows][obs_dim] = s_{i+1} - s_i (fp32 subtraction).
 * Rows >= capacity_rows are not written; n * (K - 1) rows always suffice. */
int ssc_dataset_build(const ssc_transition_log *log, int32_t obs_dim, int32_t K, int64_t n, const int32_t *d_len,
                      const int64_t *d_off, int64_t capacity_rows, float *d_X, float *d_Y, float *d_Z,
                      ssc_stream_t stream);

/* HOST function (no GPU work): path_shortcutter (smartstart/utilities/numerical.py:226-246) with the elliptical distance of
 * :116-124 -- every pair of states (i, j >= i + 2) of path[n][d] within theta of each other is a candidate shortcut, the
 * weighted interval scheduling of :189-222 (weight j - i - 1 per interval after the first, later intervals win ties) picks
 * the non-overlapping set that deletes the most interior states.  keep[i] = 1 for the states that stay, *n_kept their count
 * (may be NULL).  fp64, the reference's expression order: decisions identical to the numpy implementation. */
int ssc_path_shortcut(const double *path, int32_t n, int32_t d, const double *radii, double theta, uint8_t *keep,
                      int32_t *n_kept);

/* mean_c = mean(x[:, c]); std_c = sqrt(mean((x[:, c] - mean_c)^2)) (NND_MB_agent.py:302-304: np.mean, then
 * np.std of the centred column), accumulated in f64 in a fixed order (bit-reproducible).  rows >= 1,
 * cols <= 64. */
size_t ssc_column_stats_workspace_bytes(int32_t cols);
int ssc_column_stats(const float *d_x, int64_t rows, int32_t cols, double *d_mean, double *d_std, void *d_workspace,
                     size_t workspace_bytes, ssc_stream_t stream);

/* d_out[r][out_col0 + c] = np.nan_to_num((x[r][c] - mean_c) / std_c) (NND_MB_agent.py:305,310,315), evaluated
 * in f64 and rounded to fp32; d_out has out_stride columns -- so dataX and dataY land side by side in the
 * network-input matrix of :318 (np.concatenate((dataX, dataY), axis=1)). */
int ssc_zscore(const float *d_x, int64_t rows, int32_t cols, const double *d_mean, const double *d_std, float *d_out,
               int32_t out_stride, int32_t out_col0, ssc_stream_t stream);

/* The same for the whole network-input matrix of :318 in one pass: d_out[r] = [zscore(x[r]) | zscore(y[r])], rows of
 * cols_x + cols_y floats -- np.concatenate((dataX, dataY), axis=1) of the two z-scored matrices (NND_MB_agent.py:303-318).
 * Bit-identical to two ssc_zscore calls into the same matrix; every output line is written whole. */
int ssc_zscore_concat(const float *d_x, int32_t cols_x, const double *d_mean_x, const double *d_std_x, const float *d_y,
                      int32_t cols_y, const double *d_mean_y, const double *d_std_y, int64_t rows, float *d_out,
                      ssc_stream_t stream);

/* helper_funcs.add_noise: x[r][c] += N(0, |mean_c * noise_to_signal|) in the columns where
 * mean_c * noise_to_signal > 0 (only those, :14).  One Philox(seed; r, stream_id << 8 | (c >> 2), TAG_DATA_NOISE = 6)
 * evaluation serves four columns: word pair (x, y) for c & 3 in {0, 1}, (z, w) for {2, 3}; the cos output of the pair's
 * Box-Muller transform for the even column, the sin output for the odd one; oracle: add_noise_keyed. */
int ssc_add_noise(float *d_x, int64_t rows, int32_t cols, const double *d_mean, double noise_to_signal, uint64_t seed,
                  uint64_t stream_id, ssc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SSC_H */
