import sys, os, subprocess, glob
ROOT = "/root/repo" if os.path.exists("/root/repo") else os.getcwd()
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = sys.argv[1]
import torch, numpy as np
from smartstartcontinuous_amd import smartstart as SS
rng = np.random.default_rng(0)
s = torch.as_tensor(rng.normal(size=(100000, 2)).astype(np.float32) * [0.3, 0.02], dtype=torch.float32, device="cuda")
pts = s[:2000].clone()
wh, norm = SS.kde_scott_bandwidth(s)
for _ in range(5): SS.kde_evaluate(s, pts, wh, norm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): p = SS.kde_evaluate(s, pts, wh, norm)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1].split("/")[-1], "%%.4f ms" %% (e0.elapsed_time(e1) / 20), float(p.double().sum()))
''' % ROOT
for lib in ["smartstartcontinuous_amd/libssc.so"] + sorted(glob.glob(os.path.join(ROOT, "tools/_build/libssc_kde_*.so"))):
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, lib)], capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-300:], flush=True)
