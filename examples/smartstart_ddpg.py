#!/usr/bin/env python3
"""SmartStart around DDPG with the NND_MB navigator -- the flow of the reference's
examples/continuous/SmartStart_DDPG_Baselines_example.py:29-130 with every numeric piece on the MI355X:

  * the navigator collects its own random-rollout data set (CollectSamples, 25 x 333 steps), formats and
    z-scores it and trains its dynamics model on the device;
  * every SmartStart episode picks its start state with the critic value + Gaussian-KDE + UCB kernels, plans a
    waypoint path and follows it with MPC (sample -> forward simulation -> scoring -> argmax, one HIP path);
  * DDPG trains with the one-workgroup learner kernel.
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
    print("navigator data set: %d rows, std_x %s" % (nav.dataX.shape[0], np.round([nav.dyn_model.norm.std_x[i] for i in range(2)], 4)))
    summary = ssc.rlTrain(agent, env, print_results=True, print_steps=False, num_episodes=args.episodes,
                          max_steps=args.max_steps)
    print("smart-start episodes:", summary.smart_start_episodes, "| navigator trainings:",
          int(nav.dyn_model._adam["t"].item()) if hasattr(nav.dyn_model, "_adam") else 0, "Adam steps")
    if args.save_dir:
        os.makedirs(args.save_dir, exist_ok=True)
        print("summary written to", summary.save(args.save_dir))


if __name__ == "__main__":
    main()
