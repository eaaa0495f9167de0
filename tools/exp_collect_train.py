"""Training-loss trajectory of a 1x32 dynamics model on (a) the reference's recorded data set and (b) the data
set NND_MB_agent collects itself on the device (same env, same sizes)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smartstartcontinuous_amd as ssc  # noqa: E402
from smartstartcontinuous_amd.agents import NND_MB_agent  # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "mc_reference_rollouts.npz"))
for tag in ("reference", "collected", "collected-clean"):
    for seed in (3, 123):
        env = ssc.make("MountainCarContinuous-v0", seed=seed)
        kw = dict(training_data=dict(dataX=g["dataX"], dataY=g["dataY"], dataZ=g["dataZ"])) if tag == "reference" else \
            dict(make_training_dataset_noisy=(tag == "collected"))
        agent = NND_MB_agent(env, None, num_fc_layers=1, depth_fc_layers=32, precision="f32", seed=seed, **kw)
        nm = agent.dyn_model.norm
        line = []
        for ep in range(6):
            line.append(agent.train_dynamics_model(nEpoch=5, fraction_use_new=0.0, rng=np.random.RandomState(ep)))
        print(tag, seed, "loss every 5 epochs:", " ".join(f"{l:.4f}" for l in line),
              "| std_x", [round(nm.std_x[i], 5) for i in range(2)], "std_z", [f"{nm.std_z[i]:.3e}" for i in range(2)],
              "mean_z", [f"{nm.mean_z[i]:.3e}" for i in range(2)])
