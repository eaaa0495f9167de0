#!/usr/bin/env python3
"""The SmartStart navigator on its own, everything resident in HBM:

  1. random-policy rollouts of N envs (one fused launch) -> (s, a, s' - s) training set, statistics, z-scores;
  2. Dyn_Model.train on the device (fused one-launch steps for one hidden layer, fp32-MFMA GEMMs otherwise);
  3. P envs each follow a recorded path with MPC as the rollout policy (HIP-graph replay of the 5-launch step).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import smartstartcontinuous_amd as ssc  # noqa: E402
from smartstartcontinuous_amd import collect_samples as cs, navigator as nav, numerical as num  # noqa: E402
from smartstartcontinuous_amd.agents import init_dynamics_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rollouts", type=int, default=4096)
    ap.add_argument("--steps-per-rollout", type=int, default=333)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--problems", type=int, default=16)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=4)
    ap.add_argument("--follow-steps", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--precision", choices=["bf16_mfma", "f32"], default="bf16_mfma",
                    help="forward simulation of the MPC: bf16 MFMA kernel (default) or the fp32 parity path")
    args = ap.parse_args()
    env1 = ssc.make("MountainCarContinuous-v0", seed=args.seed)

    t0 = time.perf_counter()
    collector = cs.CollectSamples(env1, cs.Policy_Random(env1), seed=args.seed)
    ts = collector.collect_dataset(args.rollouts, args.steps_per_rollout)
    (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (ts.dataX, ts.dataY, ts.dataZ))
    inputs = torch.empty((len(ts), 3), device="cuda")
    cs.zscore_into(ts.dataX, mx, sx, inputs, 0)
    cs.zscore_into(ts.dataY, my, sy, inputs, 2)
    outputs = cs.zscore_into(ts.dataZ, mz, sz, torch.empty_like(ts.dataZ))
    torch.cuda.synchronize()
    print("data set: %d rows from %d rollouts in %.1f ms" % (len(ts), args.rollouts, (time.perf_counter() - t0) * 1e3))

    host = lambda t: t.cpu().numpy()
    norm = dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz))
    Ws, bs = init_dynamics_weights(3, 2, args.layers, args.depth, torch.Generator().manual_seed(args.seed))
    model = nav.DynamicsModel(Ws, bs, norm, 2, 1, precision=args.precision)
    t0 = time.perf_counter()
    loss = model.train(inputs, outputs, np.zeros((0, 3)), np.zeros((0, 2)), args.epochs, 0.0,
                       rng=np.random.RandomState(args.seed))
    print("trained %dx%d for %d epochs (%d steps) in %.1f ms, last-epoch loss %.4f"
          % (args.layers, args.depth, args.epochs, args.epochs * (len(ts) // 512), (time.perf_counter() - t0) * 1e3, loss))

    # every problem follows the states of one recorded validation rollout
    states, _, _, _ = collector.collect_samples(args.problems, 200)
    P = args.problems
    wps, lefts, radii = [], [], []
    for path in states:
        stds, means = num.path_deltas_stds_and_means_per_dim(path)
        rad = num.radii_calc(means, stds, 1, 1, 1)
        dist = num.elliptical_euclidean_distance_function_generator(rad)
        short = num.path_shortcutter(path, dist, 1)
        wp = np.asarray(num.get_start_waypoints_final_states_steps(short, 1))
        wps.append(wp); radii.append(rad); lefts.append(num.distances_left(wp, dist))
    problems = nav.MpcProblemSet(wps, lefts, radii, [0] * P, theta=1.0, gamma=0.75, horizontal_penalty_factor=0.5)
    batch = nav.NavigatorBatch(model, problems, num_control_samples=args.samples, horizon=args.horizon, seed=args.seed)
    venv = ssc.VecEnv("MountainCarContinuous-v0", P, seed=args.seed + 2)
    venv.reset()
    start = torch.as_tensor(np.stack([p[0] for p in states]), dtype=torch.float32, device="cuda")
    venv.s0.copy_(start[:, 0]); venv.s1.copy_(start[:, 1])
    venv.rollout(8, policy=ssc.MpcPolicy(batch))            # warm-up + graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    venv.rollout(args.follow_steps, policy=ssc.MpcPolicy(batch))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    idx = batch.problems.cur_idx.cpu().numpy() if hasattr(batch.problems, "cur_idx") else None
    print("%d envs followed their paths for %d MPC steps (%d x %d samples, horizon %d) in %.1f ms = %.3f ms per step"
          % (P, args.follow_steps, P, args.samples, args.horizon, dt * 1e3, dt * 1e3 / args.follow_steps))
    if idx is not None:
        print("waypoint reached per env:", idx.tolist(), "of", [len(w) for w in wps])


if __name__ == "__main__":
    main()
