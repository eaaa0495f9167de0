"""us per DDPG training iteration (ssc_ddpg_train_ws) over the reference's network grid x batch sizes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent


def run(h1, h2, B, n_it=200, obs_dim=2, wide=None):
    if wide is None:
        os.environ.pop("SSC_DDPG_WIDE", None)
    else:
        os.environ["SSC_DDPG_WIDE"] = "1" if wide else "0"
    rng = np.random.default_rng(0)
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=h1, actor_h2=h2, critic_h1=h1, critic_h2=h2,
                                 lastLayerTanh=True, seed=1, training=False, batch_size=B)
    cap = 100000
    dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
    s, a = dev(rng.uniform(-1.2, 0.6, (cap, obs_dim)), torch.float32), dev(rng.uniform(-1, 1, (cap, 1)), torch.float32)
    r, t = dev(rng.normal(size=cap), torch.float32), dev(rng.random(cap) < 0.01, torch.uint8)
    idx = torch.randint(0, cap, (n_it, B), dtype=torch.int32, device="cuda")
    for _ in range(2):
        agent.train_on(s, a, r, t, s, idx, n_it)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    e0.record(); agent.train_on(s, a, r, t, s, idx, n_it); e1.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n_it * 1e3
    print(json.dumps(dict(actor_critic="%d-%d" % (h1, h2), batch=B, forced_wide=wide, us_per_iteration=round(us, 2),
                          host_enqueue_us_per_iteration=round(host / n_it * 1e6, 2), samples_per_s=round(B / (us * 1e-6)))), flush=True)


if __name__ == "__main__":
    run(64, 32, 64)
    run(64, 32, 64, wide=True)
    for h1, h2 in ((64, 32), (128, 64), (200, 100)):
        for B in (64, 256, 1024, 4096):
            if (h1, h2, B) != (64, 32, 64):
                run(h1, h2, B)
