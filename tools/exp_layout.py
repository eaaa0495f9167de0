"""A/B: packed [K][cols][N] vs dense [K][N] transition chunk, interleaved rounds in one process."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smartstartcontinuous_amd import VecEnv, RandomPolicy, TransitionChunk
n, K = 65536, 1024
env = VecEnv("MountainCarContinuous-v0", n, seed=1); env.reset()
pd = env.policy_desc(RandomPolicy())
chunks = {"packed": TransitionChunk(2, K, n, env.device, packed=True), "dense": TransitionChunk(2, K, n, env.device, packed=False)}
res = {k: [] for k in chunks}
for rnd in range(12):
    for name, c in chunks.items():
        for _ in range(2): env.rollout(K, out=c, policy_desc=pd)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): env.rollout(K, out=c, policy_desc=pd)
        b.record(); torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b) / 5)
for k, v in res.items():
    v = sorted(v)
    print(k, "median %.4f ms  min %.4f  max %.4f  -> %.0f GB/s median" % (v[len(v)//2], v[0], v[-1], n*K*25/v[len(v)//2]/1e6))
