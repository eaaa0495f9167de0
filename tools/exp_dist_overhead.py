import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv
from smartstartcontinuous_amd.sharding import TransitionGather
n, K = 65536, 1024
env = VecEnv("MountainCarContinuous-v0", n, device=dev, seed=1234); env.reset()
chunks = [TransitionChunk(2, K, n, dev) for _ in range(2)]
pd = env.policy_desc(RandomPolicy())
tg = TransitionGather(2, 16, n, 1, 0, dev)
def run(mode, steps=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        c = chunks[i & 1]
        if mode != "none": tg.wait_buffer_free(i & 1)
        env.rollout(K, out=c, policy_desc=pd)
        if mode == "full": tg.submit(c, i & 1, env.stats)
        elif mode == "pack_only":
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(tg.side):
                tg.side.wait_event(ev); tg.pack(c, i & 1, env.stats); e2 = torch.cuda.Event(); e2.record(); tg.packed[i & 1] = e2
        elif mode == "gather_only":
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(tg.side):
                tg.side.wait_event(ev); dist.gather(tg.send[0], tg.recv, dst=0)
        elif mode == "allreduce_only":
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(tg.side):
                tg.side.wait_event(ev); dist.all_reduce(tg._global_stats)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"{mode:15s} host enqueue {t_host/steps*1e3:.3f} ms/step   total {t_all/steps*1e3:.3f} ms/step", flush=True)
for m in ("none", "pack_only", "gather_only", "allreduce_only", "full", "none"):
    run(m)
dist.destroy_process_group()
