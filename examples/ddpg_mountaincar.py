#!/usr/bin/env python3
"""DDPG on the edited continuous MountainCar, two ways on one MI355X:

  * ``--mode single``: the reference's own call structure (examples/continuous/DDPG_Baselines_example.py:28-80 --
    make_timed_env -> DDPG_Baselines_agent -> rlTrain -> Summary.save) with the env, the actor/critic and the
    train step running through libssc.so;
  * ``--mode vec``: the vectorised actor-learner loop -- N envs roll out under the current actor in fused
    chunks, the transitions go into a device replay ring, the learner runs on minibatches drawn from it; nothing
    but the loss scalars and the finished-episode records leaves HBM.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

import smartstartcontinuous_amd as ssc  # noqa: E402
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent  # noqa: E402


def make_agent(env, seed):
    return DDPG_Baselines_agent(env, None, buffer_size=100000, batch_size=64, num_train_iterations=50,
                                num_steps_before_train=200, ou_epsilon=1.0, ou_min_epsilon=0.01,
                                ou_epsilon_decay_factor=.99, ou_mu=0.4, ou_sigma=0.6, ou_theta=.15, actor_lr=0.001,
                                actor_h1=64, actor_h2=32, critic_lr=0.001, critic_h1=64, critic_h2=32,
                                lastLayerTanh=True, seed=seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["single", "vec"], default="vec")
    ap.add_argument("--power-scalar", type=float, default=1.0)
    ap.add_argument("--episodes", type=int, default=5, help="single: episodes to run")
    ap.add_argument("--envs", type=int, default=4096, help="vec: parallel envs")
    ap.add_argument("--chunks", type=int, default=20, help="vec: rollout chunks of 250 steps")
    ap.add_argument("--overlap", action="store_true",
                    help="vec: roll chunk i+1 on a second stream while the learner works on chunk i (one chunk stale)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--save-dir", default=None)
    args = ap.parse_args()
    np.random.seed(args.seed)
    if args.mode == "single":
        env = ssc.Continuous_MountainCarEnv_Editted.make_timed_env(args.power_scalar, max_episode_steps=1000,
                                                                   seed=args.seed)
        agent = make_agent(env, args.seed)
        summary = ssc.rlTrain(agent, env, print_results=True, print_steps=False, num_episodes=args.episodes,
                              max_steps=1000)
    else:
        env = ssc.VecEnv("MountainCarContinuousActionX%s-v0" % args.power_scalar, args.envs, seed=args.seed)
        agent = make_agent(ssc.SingleEnvView(ssc.VecEnv(env.spec.id, 1, seed=args.seed)), args.seed)
        summary, losses, replay = ssc.rl_train_vec_ddpg(env, agent, num_chunks=args.chunks, chunk_steps=250,
                                                        replay_capacity=1 << 20, train_iters=50, overlap=args.overlap)
        goals = sum(1 for steps, ret in summary.episodes if ret > 0)
        print("%d env-steps, %d finished episodes (%d reached the goal), %d records in the replay ring, "
              "last critic/actor loss %.4g / %.4g" % (args.envs * args.chunks * 250, len(summary), goals, len(replay),
                                                      *losses[-1][-1].tolist()))
    if args.save_dir:
        os.makedirs(args.save_dir, exist_ok=True)
        print("summary written to", summary.save(args.save_dir))
    print("episodes: %d, best total reward %.2f" % (len(summary), summary.best_reward))


if __name__ == "__main__":
    main()
