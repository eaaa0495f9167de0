"""What do the per-launch HIP events of bench.py cost?  After settling: 20-launch windows (sync, K launches, sync; host
clock and one bracketing event pair) with and without an event pair around EVERY launch."""
import sys, os, time, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv
n, K = 65536, 1024
env = VecEnv("MountainCarContinuous-v0", n, seed=1234); env.reset()
chunks = [TransitionChunk(env.obs_dim, K, n, env.device) for _ in range(2)]
pd = env.policy_desc(RandomPolicy())
cnt = [0]
def launch():
    env.rollout(K, out=chunks[cnt[0] & 1], policy_desc=pd); cnt[0] += 1
for _ in range(1500): launch()
torch.cuda.synchronize()
def window(per_launch, m=20):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(m)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for a, b in evs:
        if per_launch: a.record()
        launch()
        if per_launch: b.record()
    e1.record()
    torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / m * 1e3
    return host, e0.elapsed_time(e1) / m, (st.mean(a.elapsed_time(b) for a, b in evs) if per_launch else float("nan"))
for rep in range(6):
    for pl in (True, False):
        h, br, per = window(pl)
        print("per-launch events %-5s: host %.4f ms/step, bracketing events %.4f ms/step, mean of per-launch pairs %.4f" % (pl, h, br, per))
