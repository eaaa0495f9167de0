#!/usr/bin/env python3
"""The reference's navigator test (smartstart/RLAgents/NND_MB_agent_main.py) with this engine: NND_MB_agent (1x32, lr 1e-3,
30 epochs, N = 500, horizon 4, gamma .75, penalty .5, shortcutting on, retraining on the aggregated replay data every
episode) follows goal-reaching paths in stock MountainCarContinuous-v0; prints the per-episode (steps, return) next to the
band of the reference's 29 shipped goal-reaching episodes (tests/golden/nnd_mb_runs.npz).

    python tools/exp_nnd_mb_runs.py [precision=f32] [episodes=10] [seeds=3]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smartstartcontinuous_amd as ssc                                     # noqa: E402
from smartstartcontinuous_amd.agents import NND_MB_agent                   # noqa: E402


def run(precision, episodes, seed, target_path, data, max_steps=1000):
    env = ssc.make("MountainCarContinuous-v0", seed=seed)
    np.random.seed(seed)
    agent = NND_MB_agent(env, None, steps_per_waypoint=1, mean_per_stepsize=1, std_per_stepsize=1,
                         stepsizes_in_waypoint_radii=1, gamma=.75, horizontal_penalty_factor=.5, horizon=4,
                         num_control_samples=500, path_shortcutting=True, steps_before_giving_up_on_waypoint=5,
                         num_episodes_for_aggregation=1, depth_fc_layers=32, num_fc_layers=1, nEpochs=30,
                         training_data=data, precision=precision, seed=seed)
    out = []
    for ep in range(episodes):
        obs = env.reset()
        agent.start_new_episode_plan(obs, target_path)
        total, n = 0.0, 0
        for step in range(max_steps):
            action, _pred = agent.get_action_with_predicted_states(obs)
            obs2, r, done, _ = env.step(action)
            agent.observe(obs, action, r, obs2, done)
            total += r
            n += 1
            if done:
                break
            obs = obs2
        agent.end_episode()
        out.append((n, total))
    return out


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
    episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    g = np.load(os.path.join(ROOT, "tests/golden/mc_reference_rollouts.npz"))
    data = dict(dataX=g["dataX"], dataY=g["dataY"], dataZ=g["dataZ"])
    runs = np.load(os.path.join(ROOT, "tests/golden/nnd_mb_runs.npz"))
    ok = runs["steps"] < 999
    print("reference: %d goal-reaching episodes, steps %d-%d, return %.2f-%.2f" % (
        ok.sum(), runs["steps"][ok].min(), runs["steps"][ok].max(), runs["returns"][ok].min(), runs["returns"][ok].max()))
    offs = np.concatenate([[0], np.cumsum(runs["path_lens"])])
    for s in range(seeds):
        k = s % len(runs["path_lens"])
        target = runs["path_states"][offs[k]:offs[k + 1]]
        t0 = time.time()
        eps = run(precision, episodes, 1234 + s, target, data)
        print("seed %d, target path %d (%d states, return %.2f): %s  [%.1f s]" % (
            1234 + s, k, len(target), runs["path_returns"][k], [(n, round(r, 2)) for n, r in eps], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
