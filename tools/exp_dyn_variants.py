"""A/B of dyn_mfma_sim_kernel builds: each variant is a separately built libssc (tools/_build/libssc_*.so)
run in its own process on the BASELINE config-4 shape (65536 rows, 4-500-500-3, H=20).
usage: python tools/exp_dyn_variants.py [lib ...]   (default: product lib + every tools/_build/libssc_*.so)"""
import sys, os, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = sys.argv[1]
import torch, numpy as np
from exp_nav import make, timeit
dims, M, H = (4, 500, 500, 3), 65536, 20
model, d, a = make(dims)
A = torch.rand((M, H, a), device="cuda") * 2 - 1
s0 = torch.randn((M, d), device="cuda") * 0.3
S = torch.empty((H + 1, M, d), device="cuda")
fn = lambda: model.do_forward_sim(s0, A, precision="bf16_mfma", out=S)
for _ in range(20): fn()
med, mn = timeit(fn, reps=15)
flop = 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
chk = float(S[-1].double().abs().mean())
print(sys.argv[1].split("/")[-1], "median %%.4f ms min %%.4f  -> %%.0f TFLOP/s  |S_H| %%.6f" %% (med, mn, flop * M * H / med / 1e9, chk))
''' % (ROOT, ROOT)
libs = sys.argv[1:] or (["smartstartcontinuous_amd/libssc.so"] + sorted(glob.glob(os.path.join(ROOT, "tools/_build/libssc_*.so"))))
for rnd in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, lib)], capture_output=True, text=True)
        print([l for l in out.stdout.splitlines() if "median" in l] or out.stderr[-400:], flush=True)
