"""A/B of rollout block sizes: each variant is a separately built libssc (tools/_build/libssc_b<N>.so),
run in its own process; prints median/min ms per 65536 x 1024 chunk."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = sys.argv[1]
import torch
from smartstartcontinuous_amd import VecEnv, RandomPolicy, TransitionChunk
n, K = 65536, 1024
env = VecEnv("MountainCarContinuous-v0", n, seed=1); env.reset()
c = TransitionChunk(2, K, n, env.device); pd = env.policy_desc(RandomPolicy())
ts = []
for r in range(14):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): env.rollout(K, out=c, policy_desc=pd)
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
ts = sorted(ts[2:])
print(sys.argv[1].split("/")[-1], "median %%.4f min %%.4f" %% (ts[len(ts)//2], ts[0]))
''' % ROOT
for rnd in range(2):
    for lib in ["smartstartcontinuous_amd/libssc.so", "tools/_build/libssc_b64.so", "tools/_build/libssc_b128.so", "tools/_build/libssc_b512.so"]:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, lib)], capture_output=True, text=True)
        print([l for l in out.stdout.splitlines() if "median" in l] or out.stderr[-300:], flush=True)
