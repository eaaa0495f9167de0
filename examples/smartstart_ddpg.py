#!/usr/bin/env python3
"""SmartStart around DDPG with the NND_MB navigator -- the flow of the reference's
examples/continuous/SmartStart_DDPG_Baselines_example.py:29-130 with every numeric piece on the MI355X:

  * the navigator collects its own random-rollout data set (CollectSamples, 25 x 333 steps), formats and
    z-scores it and trains its dynamics model on the device;
  * every SmartStart episode picks its start state with the critic value + Gaussian-KDE + UCB kernels, plans a
    waypoint path and follows it with MPC (sample -> forward simulation -> scoring -> argmax, one HIP path);
  * DDPG trains with the one-workgroup learner kernel.

``--mode vec`` runs the same algorithm for ``--envs`` environments at once (``rl_train_vec_smartstart``): the navigator's
dynamics model is built the same way, then every env navigates / explores in its own mode inside one fused step, smart
starts are selected on the device replay ring once per chunk, and a finished env starts its next episode (with or
without a smart start) without the host.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

import smartstartcontinuous_amd as ssc  # noqa: E402
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=6)
    ap.add_argument("--max-steps", type=int, default=300)
    ap.add_argument("--power-scalar", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--mode", choices=("scalar", "vec"), default="scalar")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--chunks", type=int, default=40)
    ap.add_argument("--chunk-steps", type=int, default=64)
    ap.add_argument("--samples", type=int, default=64, help="MPC candidate sequences per env and step (vec mode)")
    ap.add_argument("--plans", type=int, default=4, help="plans on offer per selection (vec mode)")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--train-iters", type=int, default=10)
    ap.add_argument("--refresh-every", type=int, default=1, help="chunks between smart-start selections (vec mode)")
    ap.add_argument("--nav-precision", choices=("f32", "bf16_mfma"), default="f32", help="forward-simulation path of the navigator (vec mode; f32 = the fused VALU kernel for one small hidden layer)")
    ap.add_argument("--kde-max-states", type=int, default=500000, help="bound the KDE's data set by a strided subsample of the ring (vec mode; default: the reference's replay capacity; 0 = every state)")
    ap.add_argument("--sequential-selection", action="store_true", help="vec mode: select, then roll (default: the selection of a chunk overlaps its rollout and its plans go on offer one chunk later)")
    ap.add_argument("--replay-capacity", type=int, default=None, help="records in the device ring (default: two full episodes per env)")
    args = ap.parse_args()
    np.random.seed(args.seed)
    env = ssc.Continuous_MountainCarEnv_Editted.make_timed_env(args.power_scalar, max_episode_steps=args.max_steps,
                                                               seed=args.seed)
    ddpg = DDPG_Baselines_agent(env, None, buffer_size=100000, batch_size=64, num_train_iterations=50,
                                num_steps_before_train=100, ou_epsilon=1.0, ou_min_epsilon=0.01,
                                ou_epsilon_decay_factor=.99, ou_mu=0.4, ou_sigma=0.6, ou_theta=.15, actor_lr=0.001,
                                actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64, critic_h2=32,
                                lastLayerTanh=True, seed=args.seed)
    agent = ssc.SmartStartContinuous(ddpg, env, None, exploitation_param=1., exploration_param=2., eta=0.5,
                                     eta_decay_factor=1., n_ss=2000, print_ss_stuff=True,
                                     nnd_mb_final_steps=10, nnd_mb_steps_per_waypoint=1, nnd_mb_horizon=4,
                                     nnd_mb_num_control_samples=5000, nnd_mb_path_shortcutting=True,
                                     nnd_mb_num_fc_layers=1, nnd_mb_depth_fc_layers=32, nnd_mb_nEpochs=30,
                                     nnd_mb_precision="f32", nnd_mb_seed=args.seed)
    nav = agent.nnd_mb_agent
    if args.mode == "vec":
        return vec(args, nav.dyn_model)
    print("navigator data set: %d rows, std_x %s" % (nav.dataX.shape[0], np.round([nav.dyn_model.norm.std_x[i] for i in range(2)], 4)))
    summary = ssc.rlTrain(agent, env, print_results=True, print_steps=False, num_episodes=args.episodes,
                          max_steps=args.max_steps)
    print("smart-start episodes:", summary.smart_start_episodes, "| navigator trainings:",
          int(nav.dyn_model._adam["t"].item()) if hasattr(nav.dyn_model, "_adam") else 0, "Adam steps")
    if args.save_dir:
        os.makedirs(args.save_dir, exist_ok=True)
        print("summary written to", summary.save(args.save_dir))


def vec(args, dyn_model):
    import time

    import torch
    env = ssc.VecEnv("MountainCarContinuous-v0", args.envs, seed=args.seed, power_scalar=args.power_scalar,
                     max_episode_steps=args.max_steps)
    env.reset()
    ddpg = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=args.batch, num_train_iterations=args.train_iters,
                                ou_epsilon=1.0, ou_min_epsilon=0.01, ou_epsilon_decay_factor=.99, ou_mu=0.4, ou_sigma=0.6,
                                ou_theta=.15, actor_lr=0.001, actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64,
                                critic_h2=32, lastLayerTanh=True, seed=args.seed, precision="bf16_mfma")
    dyn_model.precision = args.nav_precision     # fused kernels both: bf16 MFMA (any size) or fp32 VALU (one small hidden layer)
    dyn_model.invalidate()
    smart = ssc.VecSmartStart(env, ddpg, dyn_model, eta=0.5, eta_decay_factor=1., n_ss=2000, n_plans=args.plans,
                              num_control_samples=args.samples, horizon=4, final_steps=10, chunk_steps=args.chunk_steps,
                              seed=args.seed, log_modes=True, kde_max_states=args.kde_max_states or None)
    nav_steps = []
    cap = args.replay_capacity or 2 * args.envs * args.max_steps     # a smart-start path needs its episode's start in the ring
    kw = dict(chunk_steps=args.chunk_steps, train_iters=args.train_iters, replay_capacity=cap, refresh_every=args.refresh_every,
              overlap_selection=not args.sequential_selection)
    ssc.rl_train_vec_smartstart(env, smart, 2, **kw)   # warm-up: allocations, graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    summary, losses, replay = ssc.rl_train_vec_smartstart(env, smart, args.chunks, **kw,
                                                          on_chunk=lambda c, out, sm: nav_steps.append(int(sm.mode_log.sum())))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = args.envs * args.chunk_steps * args.chunks
    eps = summary.episodes
    goals = sum(1 for (n, r) in eps if r > 0)
    print("vec smartstart: %d envs x %d steps x %d chunks in %.2f s = %.3g env-steps/s (%.2f ms per step); %d episodes, %d reached the goal, "
          "best return %.2f; %.1f %% of the env-steps were navigated; %d selections, %d plans published"
          % (args.envs, args.chunk_steps, args.chunks, dt, steps / dt, dt / (args.chunk_steps * args.chunks) * 1e3, len(eps), goals,
             summary.best_reward, 100.0 * sum(nav_steps) / steps, smart.selections, smart.pool.published))

    def timed(fn, reps=5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3
    chunk = ssc.TransitionChunk(env.obs_dim, args.chunk_steps, env.n, env.device)
    print("per chunk: selection + planning %.2f ms (of which host geometry incl. path shortcutting %.2f ms), %d-step rollout %.2f ms, "
          "replay append %.2f ms, %d learner iterations %.2f ms"
          % (timed(lambda: smart.refresh_plans(replay)),
             timed(lambda: [smart.plan_from_path(replay.get_episodic_path_to_buffer_index(replay.get_possible_smart_start_indices(1)[:1]).double().cpu().numpy()) for _ in range(args.plans)]),
             args.chunk_steps, timed(lambda: smart.rollout(args.chunk_steps, chunk)),
             timed(lambda: replay.append_chunk(chunk, reward_scale=1.0)), args.train_iters,
             timed(lambda: ddpg.train_from(replay, args.train_iters))))


if __name__ == "__main__":
    main()
