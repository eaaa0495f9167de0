#!/usr/bin/env python3
"""Fixture for the only evidence the reference holds for its NAVIGATOR end to end (dynamics model trained on
models/NND_MB_agent/default/training_data, forward simulation, MPC scoring, waypoint bookkeeping):
data/nnd_mb_tests/NND_MB_agent_MountainCarContinuous-v0*.json -- the 12 saved runs of smartstart/RLAgents/NND_MB_agent_main.py
(1x32 network, lr 1e-3, 30 epochs, N = 500 candidates, horizon 4, gamma .75, horizontal penalty .5, shortcutting on, stock
MountainCarContinuous-v0), each following a goal-reaching DDPG path.  DATA only: the per-episode (steps, total reward)
records of every run and the state paths the reference's navigator traversed in its goal-reaching episodes.

    python tests/golden/make_nnd_mb_runs.py        # in the build container; writes nnd_mb_runs.npz
"""
import glob
import json
import os

import numpy as np

REF = os.environ.get("SSC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

files = sorted(glob.glob(os.path.join(REF, "data/nnd_mb_tests/NND_MB_agent_MountainCarContinuous-v0*.json")))
run_of, steps, returns, paths, path_returns = [], [], [], [], []
for r, f in enumerate(files):
    d = json.load(open(f))
    for n, ret in d["episodes"]:
        run_of.append(r)
        steps.append(int(n))
        returns.append(float(ret))
    for p, ret in zip(d["last_paths"], d["last_rewards"]):
        p = np.asarray(p, np.float64)
        if p.ndim == 2 and 2 <= len(p) < 999:                    # a traversed goal-reaching path (placeholders have 1 state)
            paths.append(p)
            path_returns.append(float(ret))
lens = np.asarray([len(p) for p in paths], np.int32)
flat = np.concatenate(paths, axis=0)
np.savez_compressed(os.path.join(OUT, "nnd_mb_runs.npz"), run_of=np.asarray(run_of, np.int16),
                    steps=np.asarray(steps, np.int16), returns=np.asarray(returns, np.float64),
                    path_states=flat, path_lens=lens, path_returns=np.asarray(path_returns, np.float64))
ok = np.asarray(steps) < 999
print(len(files), "runs,", len(steps), "episodes,", int(ok.sum()), "reached the goal: steps",
      np.asarray(steps)[ok].min(), "-", np.asarray(steps)[ok].max(), "return",
      np.asarray(returns)[ok].min(), "-", np.asarray(returns)[ok].max(), ";", len(paths), "traversed paths")
