"""Do the FAST loops learn?  The reference's product is a learning curve (data/ddpg_baselines_summaries/good_params/*.json,
data/smart_start_continuous_summaries/...); tests/test_gpu_learning_curves.py pins the scalar ``rlTrain`` path to those
archives, this file does the same for the vectorised loops that run at 10^8 .. 10^10 env-steps/s:

  * ``rl_train_vec_ddpg`` on stock MountainCarContinuous-v0, 4096 envs, the shipped runs' hyper-parameters (network sizes,
    learning rates, gamma, tau, OU noise and its per-episode decay), learner cadence 100 x batch 64 per 32-step chunk on the
    newest 4 steps of every chunk -- the settings sweep is profiles/r04/learning/sweep.txt: all 5 seeds of this cadence reach
    the band, full-chunk appends (a replay ring that then spans only 256 env-steps) fail in 2 of 5, 25 x batch 1024 in 4 of 5;
  * ``rl_train_vec_smartstart`` at 65 536 envs: goals for 3 seeds x 2 data-set noise streams.
"""
import json
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N_ENVS, CHUNK, ITERS, BATCH, LAST, CHUNKS = 4096, 32, 100, 64, 4, 1500       # 1500 x 32 x 4096 = 2 x 10^8 env-steps
GEN_WINDOW = (90, 130)     # the reference's late window: episodes 90..129 of a run <-> epsilon 0.99^90 .. 0.99^130


@pytest.fixture(scope="module")
def ssc():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    import smartstartcontinuous_amd as pkg
    pkg._ffi.lib()
    return pkg


def test_vec_ddpg_reaches_the_reference_band(ssc, golden_dir):
    """3 seeds: the median return of the episodes that finished while the schedule was where the reference's late window
    is (generations 90..129 = the same epsilon range) lies inside the inter-decile band of the reference's 125 shipped
    runs' late-window medians (89.9 .. 93.7), nearly every one of those episodes reaches the goal, and the run gets its
    first goal within the stated env-step budget."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    g = np.load(f"{golden_dir}/ddpg_good_params_curves.npz")
    p = json.loads(str(g["param_dict"]))
    rets = g["returns"].astype(np.float64)
    band = np.percentile(np.median(rets[:, GEN_WINDOW[0]:GEN_WINDOW[1]], axis=1), [10, 90])
    assert 88.0 <= band[0] <= 92.0 and 93.0 <= band[1] <= 95.0
    meds, firsts, t0 = [], [], time.time()
    for seed in (1, 2, 3):
        env = ssc.VecEnv("MountainCarContinuous-v0", N_ENVS, seed=seed)
        env.reset()
        agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=BATCH, num_train_iterations=ITERS,
                                     ou_epsilon=p["ou_epsilon"], ou_min_epsilon=p["ou_min_epsilon"],
                                     ou_epsilon_decay_factor=p["ou_epsilon_decay_factor"], ou_mu=p["ou_mu"], ou_sigma=p["ou_sigma"],
                                     ou_theta=p["ou_theta"], actor_lr=p["actor_lr"], actor_h1=p["actor_h1"], actor_h2=p["actor_h2"],
                                     critic_lr=p["critic_lr"], critic_h1=p["critic_h1"], critic_h2=p["critic_h2"], gamma=p["gamma"],
                                     tau=p["tau"], lastLayerTanh=p["lastLayerTanh"], seed=seed)
        first = {}

        def watch(i, chunk, env_, first=first):
            if "steps" not in first and (i + 1) % 10 == 0 and float(env_.stats[1].item()) > 0:
                first["steps"] = (i + 1) * CHUNK * N_ENVS
        summary, losses, replay = ssc.rl_train_vec_ddpg(env, agent, CHUNKS, chunk_steps=CHUNK, replay_capacity=1 << 20, seed=seed,
                                                        replay_last_steps=LAST, on_chunk=watch)
        ep = np.asarray(summary.episodes, np.float64).reshape(-1, 2)
        assert summary.dropped_episode_records == 0 and len(ep) >= GEN_WINDOW[1] * N_ENVS, len(ep)
        late = ep[GEN_WINDOW[0] * N_ENVS:GEN_WINDOW[1] * N_ENVS]          # records arrive in time order (up to one drain interval)
        meds.append(float(np.median(late[:, 1])))
        firsts.append(first.get("steps"))
        eps_now = agent.decaying_ou_action_noise.epsilon
        assert abs(eps_now - max(0.99 ** (len(ep) // N_ENVS), 0.01)) < 1e-9      # one decay per generation of N_ENVS episodes
        print("seed %d: %d episodes, generations 90..129: median return %.2f, goal share %.3f, median length %.0f; first goal within %s env-steps; "
              "epsilon now %.3f (%.0f s so far)" % (seed, len(ep), meds[-1], (late[:, 0] < 999).mean(), np.median(late[:, 0]), firsts[-1],
                                                     eps_now, time.time() - t0), flush=True)
        assert (late[:, 0] < 999).mean() >= 0.95, "most late-window episodes must reach the goal"
    print("reference band", band, "ours", meds, "first goal within", firsts, "env-steps (the scalar path: episodes 0..6, i.e. < 7 000)")
    # every seed inside the band widened by one unit above (less noise than a scalar run has at the same epsilon never hurts) ...
    assert all(band[0] <= m <= band[1] + 1.0 for m in meds), (meds, band)
    # ... and the first goal within 10^7 env-steps (2 400 steps per env: the third episode)
    assert all(f is not None and f <= 10_000_000 for f in firsts), firsts


@pytest.mark.parametrize("seed,noise_stream", [(1234, 0), (1234, 1), (7, 0), (7, 1), (21, 0), (21, 1)])
def test_vec_smartstart_reaches_the_goal(ssc, seed, noise_stream):
    """65 536 envs x 20 chunks of 64 steps (TimeLimit 300: ~4 episodes per env) with the navigator built exactly as the scalar
    example builds it (its own random-rollout data set, noise from stream pair ``noise_stream``, 1 x 32 model, 30 epochs):
    smart starts are selected and navigated to, and at least 1000 episodes reach the goal -- for every seed and stream
    (round 3 had seen 0 goals for one setting of the noise stream)."""
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    n, K, chunks, max_steps = 65536, 64, 20, 300
    np.random.seed(seed)
    env1 = ssc.Continuous_MountainCarEnv_Editted.make_timed_env(1.0, max_episode_steps=max_steps, seed=seed)
    base = DDPG_Baselines_agent(env1, None, buffer_size=100000, batch_size=64, num_train_iterations=50, num_steps_before_train=100,
                                ou_mu=0.4, ou_sigma=0.6, actor_lr=0.001, actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64,
                                critic_h2=32, lastLayerTanh=True, seed=seed)
    scalar = ssc.SmartStartContinuous(base, env1, None, eta=0.5, n_ss=2000, print_ss_stuff=False, nnd_mb_horizon=4,
                                      nnd_mb_num_control_samples=5000, nnd_mb_num_fc_layers=1, nnd_mb_depth_fc_layers=32, nnd_mb_nEpochs=30,
                                      nnd_mb_precision="f32", nnd_mb_seed=seed, nnd_mb_noise_stream=noise_stream)
    model = scalar.nnd_mb_agent.dyn_model
    model.precision = "f32"
    model.invalidate()
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed, max_episode_steps=max_steps)
    env.reset()
    ddpg = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=1024, num_train_iterations=10, ou_mu=0.4,
                                ou_sigma=0.6, actor_lr=0.001, actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64, critic_h2=32,
                                lastLayerTanh=True, seed=seed, precision="bf16_mfma")
    smart = ssc.VecSmartStart(env, ddpg, model, eta=0.5, n_ss=2000, n_plans=8, num_control_samples=16, horizon=4, final_steps=10,
                              chunk_steps=K, seed=seed, log_modes=True)
    nav = []
    summary, losses, replay = ssc.rl_train_vec_smartstart(env, smart, chunks, chunk_steps=K, train_iters=10,
                                                          replay_capacity=2 * n * max_steps,
                                                          on_chunk=lambda c, out, sm: nav.append(int(sm.mode_log.sum())))
    torch.cuda.synchronize()
    ep = np.asarray(summary.episodes, np.float64).reshape(-1, 2)
    goals = int((ep[:, 1] > 0).sum())
    print("seed %d stream %d: %d episodes, %d reached the goal, best return %.2f, %.1f %% of the env-steps navigated, %d selections"
          % (seed, noise_stream, len(ep), goals, ep[:, 1].max(), 100.0 * sum(nav) / (n * K * chunks), smart.selections))
    assert len(ep) >= 3 * n and smart.selections >= chunks - 6
    assert goals >= 1000 and ep[:, 1].max() > 90.0
    assert 0.10 <= sum(nav) / (n * K * chunks) <= 0.60
