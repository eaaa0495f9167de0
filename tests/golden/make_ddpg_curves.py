#!/usr/bin/env python3
"""Fixture for the only evidence the reference holds for its DDPG path (actor forward, OU exploration, train step):
the per-episode (steps, total reward) records of its 125 shipped learning curves,
data/ddpg_baselines_summaries/good_params/*.json (1000 episodes each, stock MountainCarContinuous-v0, the canonical
hyper-parameters of SURVEY.md Appendix C).  DATA only: two [125, 1000] arrays + the hyper-parameters they were run with.

    python tests/golden/make_ddpg_curves.py        # in the build container; writes ddpg_good_params_curves.npz
"""
import glob
import json
import os

import numpy as np

REF = os.environ.get("SSC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

files = sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/good_params/*.json")))
steps, returns, params = [], [], None
for f in files:
    d = json.load(open(f))
    e = np.asarray(d["episodes"], np.float64)
    assert e.shape == (1000, 2), (f, e.shape)
    steps.append(e[:, 0].astype(np.int16))
    returns.append(e[:, 1].astype(np.float32))
    p = {k: v for k, v in d["param_dict"].items() if v != "Not serializable" and not k.startswith("zz_RANDOM")}
    assert params is None or p == params, f           # every run used the same hyper-parameters
    params = p
np.savez_compressed(os.path.join(OUT, "ddpg_good_params_curves.npz"), steps=np.stack(steps), returns=np.stack(returns),
                    param_dict=json.dumps(params, sort_keys=True))
print(len(files), "runs;", params)
