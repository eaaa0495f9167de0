"""Host time per bench step (the launch path from Python): rollout alone, rollout + gather.submit (1-rank RCCL)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv
from smartstartcontinuous_amd.sharding import TransitionGather
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, K = 65536, 1024
env = VecEnv("MountainCarContinuous-v0", n, seed=1234); env.reset()
chunks = [TransitionChunk(2, K, n, dev) for _ in range(2)]
pd = env.policy_desc(RandomPolicy())
g = TransitionGather(2, 16, n, 1, 0, dev)
def loop(reps, with_gather, K_=K):
    torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
    for i in range(reps):
        h0 = time.perf_counter()
        env.rollout(K_, out=chunks[i & 1] if K_ == K else None, policy_desc=pd, log=(K_ == K))
        if with_gather: g.submit(chunks[i & 1], i & 1, env.stats)
        host += time.perf_counter() - h0
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return host / reps * 1e6, el / reps * 1e6
for wg in (False, True, False, True):
    loop(50, wg)
    h, e = loop(300, wg)
    print("gather=%s: host %.1f us per step, wall %.1f us per step" % (wg, h, e), flush=True)
h, e = loop(300, False, 1)
print("tiny rollout (K=1, no log): host %.1f us per call, wall %.1f" % (h, e))
dist.destroy_process_group()
