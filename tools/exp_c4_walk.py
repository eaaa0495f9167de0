"""Simulation kernel (2 x 500, Pendulum dims) with more row tiles than CUs -- BASELINE configs[3] as written: 65 536 envs x 16
candidates = 1 Mi rows, H = 4 -- and the 65 536-row launch at H = 4 / 20, for the library given as argv[1] (A/B of the
walking kernel against the one-block-per-tile kernel of round 3: tools/gpu_r04_b.sh).  Sustained: every figure is the
median of launches AFTER ~0.4 s of the same launch back to back (the post-idle clock transient)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import smartstartcontinuous_amd._ffi as F
if len(sys.argv) > 1 and sys.argv[1] != "-":
    F.LIB_PATH = os.path.join(ROOT, sys.argv[1])
import numpy as np, torch
from exp_nav import make
from smartstartcontinuous_amd import navigator as nav
model, d, a = make((4, 500, 500, 3))
model.precision = "bf16_mfma"
flop_row = 2.0 * (4 * 500 + 500 * 500 + 500 * 3)
out = {"lib": os.path.basename(F.LIB_PATH)}
ONLY = sys.argv[2] if len(sys.argv) > 2 else None     # one case alone (profiling runs: rocprofv3 averages by kernel name)
for name, P, N, H in (("rows_1Mi_h4", 65536, 16, 4), ("rows_64Ki_h4", 16, 4096, 4), ("rows_64Ki_h20", 16, 4096, 20), ("rows_1Mi_h1", 65536, 16, 1)):
    if ONLY is not None and name != ONLY:
        continue
    M = P * N
    s0 = torch.randn((P, d), device="cuda") * 0.3
    S = torch.empty((H + 1, M, d), device="cuda")
    sp = nav.mpc_sampling(N, [-1.0] * a, [1.0] * a, 1234, 0, 0)
    run = lambda: model.do_forward_sim_sampled(s0, sp, M, H, precision="bf16_mfma", out=S)
    t0 = time.time()
    while time.time() - t0 < 0.4:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for e0, e1 in ev:
        e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)
    med = ms[len(ms) // 2]
    out[name] = {"ms_median": med, "ms_min": ms[0], "frac": flop_row * M * H / (med * 1e-3) / 2.5e15, "finite": bool(torch.isfinite(S).all())}
print(json.dumps(out), flush=True)
