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

# ---- the hidden-layer-size experiment (data/ddpg_baselines_summaries/hidden_layer_size_experiment/): 180 runs over
# actor / critic in {64-32, 128-64, 200-100} x learning rates {1e-3, 5e-3}, 5 runs per combination -- the evidence the
# reference holds for the WIDER networks (multi-workgroup learner, register- / LDS-staged MFMA actors)
import re  # noqa: E402

files = sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/hidden_layer_size_experiment/*.json")))
steps, returns, labels = [], [], []
for f in files:
    m = re.search(r"_a-(\d+)-(\d+)\.0-0d(\d+)_c-(\d+)-(\d+)\.0-0d(\d+)_", os.path.basename(f))
    d = json.load(open(f))
    e = np.asarray(d["episodes"], np.float64)
    assert e.shape == (1000, 2), (f, e.shape)
    p = d["param_dict"]
    a1, a2, alr, c1, c2, clr = (int(x) for x in m.groups())
    assert (p["actor_h1"], p["actor_h2"], p["critic_h1"], p["critic_h2"]) == (a1, a2, c1, c2), f
    assert abs(p["actor_lr"] - alr * 1e-3) < 1e-12 and abs(p["critic_lr"] - clr * 1e-3) < 1e-12, f
    steps.append(e[:, 0].astype(np.int16))
    returns.append(e[:, 1].astype(np.float32))
    labels.append([a1, a2, alr, c1, c2, clr])          # learning rates in units of 1e-3
np.savez_compressed(os.path.join(OUT, "ddpg_hidden_layer_curves.npz"), steps=np.stack(steps), returns=np.stack(returns),
                    labels=np.asarray(labels, np.int32))
print(len(files), "hidden-layer-size runs")

# ---- the base agent alone on the EDITED env at power_scalar 0.4 (data/ddpg_baselines_summaries/good_params_cont_mc_editted/):
# 12 runs x 1000 episodes of 1000 steps none of which ever reaches the goal -- what they record is how fast the learner
# quiets the policy down (the -0.1 a^2 action cost is the only reward signal): median return -12 around episode 80, -1.7
# around episode 180, -0.2 around episode 280.  A sharp, goal-free signature of the train step + OU exploration.
files = sorted(glob.glob(os.path.join(REF, "data/ddpg_baselines_summaries/good_params_cont_mc_editted/*.json")))
steps, returns = [], []
for f in files:
    d = json.load(open(f))
    e = np.asarray(d["episodes"], np.float64)
    assert e.shape == (1000, 2) and "ActionX0.4" in d["name"], (f, e.shape, d["name"])
    steps.append(e[:, 0].astype(np.int16))
    returns.append(e[:, 1].astype(np.float32))
np.savez_compressed(os.path.join(OUT, "ddpg_edited_env_curves.npz"), steps=np.stack(steps), returns=np.stack(returns))
print(len(files), "edited-env runs; goals:", int((np.stack(steps) < 1000).sum()))
