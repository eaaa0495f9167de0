"""Cycle stamps per level of ddpg_wide_grad_kernel (block 0): SSC_LIB_PATH=tools/_build/libssc_widediag.so python3 tools/exp_ddpg_wide_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent

for h1, h2, B in ((64, 32, 64), (200, 100, 64), (200, 100, 1024)):
    os.environ["SSC_DDPG_WIDE"] = "1"
    rng = np.random.default_rng(0)
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=h1, actor_h2=h2, critic_h1=h1, critic_h2=h2,
                                 lastLayerTanh=True, seed=1, training=False, batch_size=B)
    cap = 100000
    dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
    s, a = dev(rng.uniform(-1.2, 0.6, (cap, 2)), torch.float32), dev(rng.uniform(-1, 1, (cap, 1)), torch.float32)
    r, t = dev(rng.normal(size=cap), torch.float32), dev(rng.random(cap) < 0.01, torch.uint8)
    idx = torch.randint(0, cap, (50, B), dtype=torch.int32, device="cuda")
    agent.train_on(s, a, r, t, s, idx, 50)
    torch.cuda.synchronize()
    ws = agent._train_ws
    nb = (B + 15) // 16
    n_params = agent.actor_flat.numel() + agent.critic_flat.numel()
    off = ((nb * n_params * 4 + 255) & ~255) + 64 * 4
    st = ws[off:off + 8 * 24].view(torch.int64).cpu().numpy()
    n = int(st[0]); v = st[1:1 + n]
    d = np.diff(v) / 100.0      # s_memtime ticks at 100 MHz -> us
    print("%d-%d batch %d: entry->gather %.2f us; levels %s; wgrad %.2f; total %.2f us" %
          (h1, h2, B, d[0], " ".join("%.2f" % x for x in d[1:-1]), d[-1], (v[-1] - v[0]) / 100.0), flush=True)
