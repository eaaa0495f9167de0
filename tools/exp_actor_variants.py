"""A/B of the fused actor rollout (BASELINE config 3 shape) across separately built libraries
(default: the product library + every tools/_build/libssc_act*.so), interleaved, one process each."""
import sys, os, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = sys.argv[1]
import torch
from smartstartcontinuous_amd import VecEnv, ActorPolicy, TransitionChunk
from smartstartcontinuous_amd.agents import init_actor_weights
n, K = 65536, 256
env = VecEnv("MountainCarContinuous-v0", n, seed=1); env.reset()
w = {k: v.cuda() for k, v in init_actor_weights(2, 64, 32, 1, torch.Generator().manual_seed(1234)).items()}
c = TransitionChunk(2, K, n, env.device); pd = env.policy_desc(ActorPolicy(w))
ts = []
for r in range(14):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): env.rollout(K, out=c, policy_desc=pd)
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
ts = sorted(ts[2:])
print(sys.argv[1].split("/")[-1], "median %%.4f ms min %%.4f -> %%.3e env-steps/s" %% (ts[len(ts)//2], ts[0], n * K / ts[len(ts)//2] * 1e3))
''' % ROOT
libs = sys.argv[1:] or (["smartstartcontinuous_amd/libssc.so"] + sorted(glob.glob(os.path.join(ROOT, "tools/_build/libssc_act*.so"))))
for rnd in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, lib)], capture_output=True, text=True)
        print([l for l in out.stdout.splitlines() if "median" in l] or out.stderr[-400:], flush=True)
