"""Where do the ~50 us per step go when the bounded gather rides along (1-rank RCCL group)?  HIP-event timeline of both
streams for a few settled steps: main = rollout, side = pack + collective."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv
from smartstartcontinuous_amd.sharding import TransitionGather
n, K, G = 65536, 1024, int(os.environ.get("G", "16"))
env = VecEnv("MountainCarContinuous-v0", n, device=dev, seed=1); env.reset()
chunks = [TransitionChunk(2, K, n, dev) for _ in range(2)]
pd = env.policy_desc(RandomPolicy())
g = TransitionGather(2, G, n, 1, 0, dev)
mode = os.environ.get("MODE", "full")      # full | pack_only | none
def step(i, marks=None):
    E = lambda: torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    g.wait_buffer_free(i & 1)
    if marks is not None: a = E(); a.record(main)
    env.rollout(K, out=chunks[i & 1], policy_desc=pd)
    if marks is not None: b = E(); b.record(main)
    if mode == "full":
        g.submit(chunks[i & 1], i & 1, env.stats)
    elif mode == "pack_only":
        g.pack(chunks[i & 1], i & 1, env.stats)
    elif mode in ("events_only", "side_copy", "side_noop"):
        slot = i & 1
        if g.packed[slot] is not None: main.wait_event(g.packed[slot])
        g.pack(chunks[slot], slot, env.stats)
        ready = torch.cuda.Event(); ready.record(main)
        with torch.cuda.stream(g.side):
            g.side.wait_event(ready)
            if mode == "side_copy": g.recv[slot][0].copy_(g.send[slot], non_blocking=True)
            sent = torch.cuda.Event(); sent.record(g.side)
        if mode != "side_noop": g.packed[slot] = sent
    if marks is not None: marks.append((a, b))
for i in range(1500): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1500, 1600): step(i)
g.finish(); torch.cuda.synchronize()
print("mode %s G %d: %.1f us per step (host clock, 100 steps)" % (mode, G, (time.perf_counter() - t0) * 1e4))
marks = []
for i in range(1600, 1612): step(i, marks)
g.finish(); torch.cuda.synchronize()
base = marks[0][0]
print("rollout start / end (us since first):", " | ".join("%.0f-%.0f" % (base.elapsed_time(a) * 1e3, base.elapsed_time(b) * 1e3) for a, b in marks))
dist.destroy_process_group()
