/* Host-side sanitizer job ("host C++ built with -fsanitize=address,undefined"): libssc.so's host code
 * -- argument validation, descriptor handling, the thread-local error buffer -- built with ASan + UBSan
 * (tests/sanitizer/build.py) and driven through every entry point with invalid, empty and boundary arguments.  All
 * calls return before the first HIP call, so this runs without a GPU.  Exit code 0 and no sanitizer report = pass. */
#include <stdio.h>
#include <string.h>
#include "ssc.h"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { printf("FAILED line %d: %s  (last error: %s)\n", __LINE__, #cond, ssc_last_error()); ++failures; } } while (0)

int main(void) {
    EXPECT(ssc_version() == SSC_VERSION);
    EXPECT(ssc_last_error() != NULL);
    ssc_env_params mc, pd;
    EXPECT(ssc_env_params_default(SSC_ENV_MOUNTAINCAR, 1.0f, 999, &mc) == SSC_OK && mc.kind == SSC_ENV_MOUNTAINCAR);
    EXPECT(ssc_env_params_default(SSC_ENV_PENDULUM, 1.0f, 200, &pd) == SSC_OK && pd.max_episode_steps == 200);
    EXPECT(ssc_env_params_default(7, 1.0f, 0, &mc) == SSC_EINVAL && strlen(ssc_last_error()) > 0);
    EXPECT(ssc_env_params_default(SSC_ENV_MOUNTAINCAR, 1.0f, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_env_params_default(SSC_ENV_MOUNTAINCAR, 1.0f, 999, &mc) == SSC_OK);

    /* env step / reset / observe */
    EXPECT(ssc_mc_step(NULL, 4, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mc_step(&mc, -1, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mc_step(&mc, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SSC_OK);
    EXPECT(ssc_mc_step(&mc, 4, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mc_step(&pd, 4, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_pend_step(&pd, 4, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_pend_step(&pd, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == SSC_OK);
    EXPECT(ssc_env_reset(&mc, 4, NULL, NULL, NULL, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_env_reset(&mc, 0, NULL, NULL, NULL, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_OK);
    EXPECT(ssc_env_observe(&mc, 4, NULL, NULL, NULL, NULL) == SSC_EINVAL);

    /* fused rollout */
    ssc_policy_desc pol; memset(&pol, 0, sizeof pol);
    pol.kind = SSC_POLICY_RANDOM; pol.act_low = -1.0f; pol.act_high = 1.0f;
    ssc_rollout_state st; memset(&st, 0, sizeof st);
    EXPECT(ssc_rollout(NULL, &pol, 4, 4, &st, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_rollout(&mc, &pol, 4, 4, &st, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_EINVAL);       /* NULL state columns */
    EXPECT(ssc_rollout(&mc, &pol, 0, 4, &st, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_OK);
    EXPECT(ssc_rollout(&mc, &pol, -4, 4, &st, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_rollout(&mc, &pol, ((int64_t)1 << 31), 4, &st, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_EINVAL);
    pol.kind = 9;
    EXPECT(ssc_rollout(&mc, &pol, 4, 4, &st, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_EINVAL);
    pol.kind = SSC_POLICY_RANDOM; pol.act_low = 2.0f;
    float dummy[8] = {0}; int32_t idummy[8] = {0};
    st.s0 = dummy; st.s1 = dummy; st.steps = idummy; st.ep_ret = dummy;
    EXPECT(ssc_rollout(&mc, &pol, 4, 4, &st, NULL, NULL, NULL, 1, 0, 0, NULL) == SSC_EINVAL);       /* act_low > act_high */
    EXPECT(ssc_pack_bytes(2, 3, 5) == (size_t)(((3 * 5 * 25 + 7) & ~7) + 32));
    EXPECT(ssc_pack_transitions(NULL, 2, 4, 2, 8, NULL, NULL, NULL) == SSC_EINVAL);

    /* actor / critic */
    ssc_actor_desc ad; memset(&ad, 0, sizeof ad);
    EXPECT(ssc_actor_forward(NULL, 4, NULL, NULL, NULL) == SSC_EINVAL);
    ad.obs_dim = 2; ad.h1 = 64; ad.h2 = 32; ad.act_dim = 1;
    EXPECT(ssc_actor_forward(&ad, 4, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_actor_forward(&ad, 0, NULL, NULL, NULL) == SSC_OK);
    ad.obs_dim = 99;
    EXPECT(ssc_actor_forward(&ad, 4, NULL, NULL, NULL) == SSC_EINVAL);
    ssc_critic_desc cd; memset(&cd, 0, sizeof cd);
    EXPECT(ssc_critic_forward(NULL, 4, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    cd.obs_dim = 2; cd.act_dim = 1; cd.h1 = 64; cd.h2 = 32;
    EXPECT(ssc_critic_forward(&cd, 4, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_critic_forward(&cd, 0, NULL, NULL, NULL, NULL) == SSC_OK);

    /* dynamics model / MPC */
    ssc_mlp_desc mlp; memset(&mlp, 0, sizeof mlp);
    mlp.n_layers = 3; mlp.dims[0] = 4; mlp.dims[1] = 500; mlp.dims[2] = 500; mlp.dims[3] = 3;
    EXPECT(ssc_mlp_workspace_bytes(&mlp, 1024, SSC_PREC_F32) > 0);
    EXPECT(ssc_dyn_workspace_bytes(&mlp, 1024, SSC_PREC_BF16_MFMA) > 0);
    EXPECT(ssc_mlp_forward(&mlp, 8, NULL, NULL, SSC_PREC_F32, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_mlp_forward(NULL, 8, NULL, NULL, SSC_PREC_F32, NULL, 0, NULL) == SSC_EINVAL);
    ssc_norm nm; memset(&nm, 0, sizeof nm);
    EXPECT(ssc_dyn_forward_sim(&mlp, &nm, 8, 4, 3, 1, NULL, 1, NULL, NULL, SSC_PREC_F32, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_dyn_forward_sim(&mlp, NULL, 8, 4, 3, 1, NULL, 1, NULL, NULL, SSC_PREC_F32, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_dyn_prepare(&mlp, &nm, NULL, 0, NULL) == SSC_EINVAL);
    mlp.n_layers = 77;
    EXPECT(ssc_mlp_forward(&mlp, 8, NULL, NULL, SSC_PREC_F32, NULL, 0, NULL) == SSC_EINVAL);
    float lo[4] = {-1, -1, -1, -1}, hi[4] = {1, 1, 1, 1};
    EXPECT(ssc_mpc_sample_actions(2, 8, 4, 0, lo, hi, 1, 0, 0, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mpc_sample_actions(2, 8, 4, 1, NULL, hi, 1, 0, 0, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mpc_sample_actions(0, 8, 4, 1, lo, hi, 1, 0, 0, NULL, NULL, NULL) == SSC_OK);
    EXPECT(ssc_mpc_sample_actions(2, 8, 4, 1, lo, hi, 1, 0, 0, NULL, NULL, NULL) == SSC_EINVAL);
    ssc_mpc_problems pr; memset(&pr, 0, sizeof pr);
    EXPECT(ssc_mpc_score_workspace_bytes(4, 1000, 4) >= 256);
    EXPECT(ssc_mpc_score(NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) == SSC_EINVAL);
    pr.n_problems = 2; pr.n_samples = 8; pr.horizon = 99; pr.state_dim = 2;
    EXPECT(ssc_mpc_score(&pr, NULL, NULL, NULL, NULL, NULL, 0, NULL) == SSC_EINVAL);
    pr.horizon = 4;
    EXPECT(ssc_mpc_score(&pr, NULL, NULL, NULL, NULL, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_mpc_observe(&pr, NULL, NULL, NULL, 5, 10, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mpc_select_action(2, 8, 4, 2, 1, NULL, NULL, NULL, 0.005f, 1, 0, 0, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mpc_select_action(0, 8, 4, 2, 1, NULL, NULL, NULL, 0.005f, 1, 0, 0, NULL, NULL, NULL) == SSC_OK);
    ssc_mpc_nav_state nav; memset(&nav, 0, sizeof nav);
    EXPECT(ssc_mpc_rollout_step(&mc, &pr, &nav, NULL, NULL, 0.0f, 1, 0, &st, NULL, NULL, NULL, 1, 0, NULL, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_mpc_rollout_step(NULL, &pr, &nav, NULL, NULL, 0.0f, 1, 0, &st, NULL, NULL, NULL, 1, 0, NULL, NULL, NULL, NULL, NULL) == SSC_EINVAL);

    /* SmartStart selection, replay ring */
    float wh[4] = {1, 0, 0, 1};
    EXPECT(ssc_kde_evaluate(2, 10, NULL, 5, NULL, wh, 1.0, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_kde_evaluate(99, 10, NULL, 5, NULL, wh, 1.0, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_ucb_argmax(5, NULL, NULL, 1.0f, 2.0f, 100.0, 1.0, NULL, NULL, NULL) == SSC_EINVAL);
    ssc_replay_ring ring; memset(&ring, 0, sizeof ring);
    ssc_transition_log log; memset(&log, 0, sizeof log);
    EXPECT(ssc_replay_append(NULL, &log, 4, 8, 0, 1.0f, NULL) == SSC_EINVAL);
    ring.capacity = 100; ring.obs_dim = 2; ring.act_dim = 1;
    EXPECT(ssc_replay_append(&ring, &log, 4, 8, 0, 1.0f, NULL) == SSC_EINVAL);
    EXPECT(ssc_replay_append(&ring, &log, 0, 8, 0, 1.0f, NULL) == SSC_OK);
    ring.act_dim = 3;
    EXPECT(ssc_replay_append(&ring, &log, 4, 8, 0, 1.0f, NULL) == SSC_EINVAL);
    ring.act_dim = 1;
    EXPECT(ssc_replay_sample(1, 0, 10, 4, 65, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_replay_sample(1, 0, 10, 4, 64, NULL, NULL) == SSC_EINVAL);      /* 10 records < 64 */
    EXPECT(ssc_replay_sample(1, 0, 100, 0, 64, NULL, NULL) == SSC_OK);
    EXPECT(ssc_replay_smart_start_workspace_bytes(2000) >= 8 * 8192 && ssc_replay_smart_start_workspace_bytes(0) == 256);
    int32_t n_out = 0;
    EXPECT(ssc_replay_smart_start_indices(&ring, 50, 4, 10, 1, 0, NULL, &n_out, NULL, 0, NULL) == SSC_EINVAL);   /* no index kept */
    ring.ep_steps = idummy; ring.ep_run = idummy;
    EXPECT(ssc_replay_smart_start_indices(&ring, 50, 4, 5000, 1, 0, NULL, &n_out, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_replay_smart_start_indices(&ring, 50, 4, 10, 1, 0, NULL, NULL, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_replay_smart_start_indices(&ring, 50, 0, 10, 1, 0, NULL, &n_out, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_replay_episode_path(&ring, 50, 4, NULL, 100, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_replay_episode_path(&ring, 50, 4, idummy, 0, dummy, idummy, NULL) == SSC_EINVAL);

    /* learners */
    ssc_ddpg_desc dd; memset(&dd, 0, sizeof dd);
    ssc_replay_view rv; memset(&rv, 0, sizeof rv);
    EXPECT(ssc_ddpg_train(NULL, &rv, NULL, 1, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_ddpg_train(&dd, &rv, NULL, 1, NULL, NULL) == SSC_EINVAL || ssc_ddpg_train(&dd, &rv, NULL, 1, NULL, NULL) == SSC_EUNSUPPORTED);
    ssc_mlp_train_desc td; memset(&td, 0, sizeof td);
    td.n_layers = 2; td.dims[0] = 3; td.dims[1] = 32; td.dims[2] = 2;
    EXPECT(ssc_mlp_train_workspace_bytes(&td, 512) > 0 && ssc_mlp_train_workspace_bytes(NULL, 512) == 0);
    EXPECT(ssc_mlp_train_steps(&td, NULL, NULL, NULL, 512, 3, NULL, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_mlp_train_step(&td, NULL, NULL, NULL, 0, NULL, NULL, 0, NULL) == SSC_EINVAL);
    td.n_layers = 9;
    EXPECT(ssc_mlp_train_steps(&td, NULL, NULL, NULL, 512, 1, NULL, NULL, 0, NULL) == SSC_EINVAL);

    /* data-set kernels */
    EXPECT(ssc_dataset_scan_workspace_bytes(65536) == (65536 / 64 + 1) * 8 && ssc_dataset_scan_workspace_bytes(-1) == 0);
    EXPECT(ssc_dataset_scan(NULL, 4, 8, NULL, NULL, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_dataset_scan(&log, -1, 8, NULL, NULL, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_dataset_build(&log, 0, 4, 8, NULL, NULL, 10, NULL, NULL, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_dataset_build(&log, 2, 1, 8, NULL, NULL, 10, NULL, NULL, NULL, NULL) == SSC_OK);
    EXPECT(ssc_column_stats_workspace_bytes(3) == (size_t)2048 * 3 * 8 && ssc_column_stats_workspace_bytes(65) == 0);
    EXPECT(ssc_column_stats(NULL, 10, 65, NULL, NULL, NULL, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_zscore(NULL, 5, 3, NULL, NULL, NULL, 2, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_zscore(NULL, 0, 3, NULL, NULL, NULL, 4, 1, NULL) == SSC_OK);
    EXPECT(ssc_add_noise(NULL, 5, 300, NULL, 0.01, 1, 0, NULL) == SSC_EINVAL);
    EXPECT(ssc_add_noise(NULL, 0, 3, NULL, 0.01, 1, 0, NULL) == SSC_OK);

    EXPECT(ssc_zscore_concat(NULL, 0, NULL, NULL, NULL, 1, NULL, NULL, 5, NULL, NULL) == SSC_EINVAL);
    EXPECT(ssc_zscore_concat(NULL, 2, NULL, NULL, NULL, 1, NULL, NULL, 0, NULL, NULL) == SSC_OK);
    /* host-only geometry: runs to completion under the sanitizers.  Five states on a line, theta 1: (1, 3) and (2, 4) are
     * the candidate shortcuts (states 1 and 3 coincide, so do 2 and 4 within theta); one of them is taken */
    {
        const double path5[10] = {0, 0, 1, 1, 2, 2, 1, 1, 2, 2}, radii2[2] = {1, 1}, bad[2] = {1, 0};
        uint8_t keep5[5];
        int32_t kept = -1;
        EXPECT(ssc_path_shortcut(path5, 5, 2, radii2, 1.0, keep5, &kept) == SSC_OK && kept == 4 && keep5[0] && keep5[4]);
        EXPECT(ssc_path_shortcut(path5, 0, 2, radii2, 1.0, keep5, &kept) == SSC_OK && kept == 0);
        EXPECT(ssc_path_shortcut(path5, 5, 2, bad, 1.0, keep5, NULL) == SSC_EINVAL);
        EXPECT(ssc_path_shortcut(NULL, 5, 2, radii2, 1.0, NULL, NULL) == SSC_EINVAL);
        EXPECT(ssc_path_shortcut(path5, 5, 9, radii2, 1.0, keep5, NULL) == SSC_EINVAL);
    }

    /* the error text survives until the next failing call of this thread */
    EXPECT(ssc_add_noise(NULL, 5, 300, NULL, 0.01, 1, 0, NULL) == SSC_EINVAL && strstr(ssc_last_error(), "ssc_add_noise") != NULL);
    if (failures) { printf("%d expectation(s) failed\n", failures); return 1; }
    printf("abi_args: all expectations hold\n");
    return 0;
}
