"""Do the vectorised loops LEARN?  

  ddpg  <envs> <chunk> <iters> <batch> <chunks> <seed>     rl_train_vec_ddpg on stock MountainCarContinuous-v0, canonical
        hyper-parameters (tests/golden/ddpg_good_params_curves.npz -> param_dict): prints per window of chunks the median
        return / length of the episodes finished in it, the share that reached the goal, epsilon, and env-steps so far.
  ss    <envs> <samples> <chunks> <seed> <noise_stream>      rl_train_vec_smartstart (as examples/smartstart_ddpg.py --mode vec)
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent


def ddpg(envs, K, iters, batch, chunks, seed, last_steps=0, capacity_log2=20, window=50, quiet=0):
    g = np.load(os.path.join(ROOT, "tests", "golden", "ddpg_good_params_curves.npz"))
    p = json.loads(str(g["param_dict"]))
    env = ssc.VecEnv("MountainCarContinuous-v0", envs, seed=seed)
    env.reset()
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=batch, num_train_iterations=iters,
                                 ou_epsilon=p["ou_epsilon"], ou_min_epsilon=p["ou_min_epsilon"], ou_epsilon_decay_factor=p["ou_epsilon_decay_factor"],
                                 ou_mu=p["ou_mu"], ou_sigma=p["ou_sigma"], ou_theta=p["ou_theta"], actor_lr=p["actor_lr"], actor_h1=p["actor_h1"],
                                 actor_h2=p["actor_h2"], critic_lr=p["critic_lr"], critic_h1=p["critic_h1"], critic_h2=p["critic_h2"],
                                 gamma=p["gamma"], tau=p["tau"], lastLayerTanh=p["lastLayerTanh"], seed=seed)
    t0 = time.time()
    state = dict(first=None, last=np.zeros(4))

    def progress(i, chunk, env_):
        if (i + 1) % window:
            return
        st = env_.stats.cpu().numpy()
        d = st - state["last"]
        state["last"] = st
        if state["first"] is None and st[1] > 0:
            state["first"] = (i + 1) * K * envs
        if not quiet:
            print("chunks %4d  env-steps %.3g  episodes %7d  goal share %.3f  mean return %7.2f  mean length %5.0f  epsilon(dev) %.3f  (%.1f s)"
                  % (i + 1, st[2], d[3], d[1] / max(d[3], 1), d[0] / max(d[3], 1), d[2] / max(d[3], 1), float(agent.d_epsilon.item()), time.time() - t0), flush=True)
    s, losses, replay = ssc.rl_train_vec_ddpg(env, agent, chunks, chunk_steps=K, replay_capacity=1 << capacity_log2, seed=seed, on_chunk=progress,
                                              replay_last_steps=last_steps or None)
    ep = np.asarray(s.episodes, np.float64).reshape(-1, 2)
    late = ep[-max(1, len(ep) // 10):]
    print("cfg envs %d K %d iters %d batch %d chunks %d seed %d last %d cap 2^%d:" % (envs, K, iters, batch, chunks, seed, last_steps, capacity_log2), end=" ")
    print("first goal within %s env-steps; %d episodes; late tenth: median return %.2f, median length %.0f, goal share %.3f; dropped records %d"
          % (state["first"], len(ep), np.median(late[:, 1]), np.median(late[:, 0]), (late[:, 0] < 999).mean(), s.dropped_episode_records))


def smartstart(envs, samples, chunks, seed, noise_stream, K=64, max_steps=300):
    np.random.seed(seed)
    env1 = ssc.Continuous_MountainCarEnv_Editted.make_timed_env(1.0, max_episode_steps=max_steps, seed=seed)
    base = DDPG_Baselines_agent(env1, None, buffer_size=100000, batch_size=64, num_train_iterations=50, num_steps_before_train=100,
                                ou_epsilon=1.0, ou_min_epsilon=0.01, ou_epsilon_decay_factor=.99, ou_mu=0.4, ou_sigma=0.6, ou_theta=.15,
                                actor_lr=0.001, actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=seed)
    scalar = ssc.SmartStartContinuous(base, env1, None, eta=0.5, n_ss=2000, print_ss_stuff=False, nnd_mb_horizon=4, nnd_mb_num_control_samples=5000,
                                      nnd_mb_num_fc_layers=1, nnd_mb_depth_fc_layers=32, nnd_mb_nEpochs=30, nnd_mb_precision="f32",
                                      nnd_mb_seed=seed, nnd_mb_noise_stream=noise_stream)
    model = scalar.nnd_mb_agent.dyn_model
    env = ssc.VecEnv("MountainCarContinuous-v0", envs, seed=seed, max_episode_steps=max_steps)
    env.reset()
    ddpg_ = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=1024, num_train_iterations=10, ou_epsilon=1.0,
                                 ou_min_epsilon=0.01, ou_epsilon_decay_factor=.99, ou_mu=0.4, ou_sigma=0.6, ou_theta=.15, actor_lr=0.001,
                                 actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=seed,
                                 precision="bf16_mfma")
    model.precision = "f32"
    model.invalidate()
    smart = ssc.VecSmartStart(env, ddpg_, model, eta=0.5, n_ss=2000, n_plans=8, num_control_samples=samples, horizon=4, final_steps=10,
                              chunk_steps=K, seed=seed, log_modes=True)
    nav = []
    t0 = time.time()
    s, losses, replay = ssc.rl_train_vec_smartstart(env, smart, chunks, chunk_steps=K, train_iters=10, replay_capacity=2 * envs * max_steps,
                                                    on_chunk=lambda c, out, sm: nav.append(int(sm.mode_log.sum())))
    torch.cuda.synchronize()
    ep = np.asarray(s.episodes, np.float64).reshape(-1, 2)
    goals = int((ep[:, 1] > 0).sum())
    print("seed %d stream %d: %d envs x %d chunks x %d steps in %.1f s: %d episodes, %d reached the goal (%.4f), best return %.2f, navigated share %.3f"
          % (seed, noise_stream, envs, chunks, K, time.time() - t0, len(ep), goals, goals / max(1, len(ep)), ep[:, 1].max(), sum(nav) / (envs * K * chunks)), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a[0] == "ddpg":
        ddpg(*[int(x) for x in a[1:]])
    else:
        smartstart(*[int(x) for x in a[1:6]])
