"""A/B of ddpg_train_kernel workgroup sizes (tools/_build/libssc_ddpg<T>.so built with -DSSC_DDPG_THREADS=T)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
import smartstartcontinuous_amd._ffi as F
F.LIB_PATH = sys.argv[1]
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
rng = np.random.default_rng(0)
agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=1, training=False)
cap = 100000
dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()
s = dev(rng.uniform(-1.2, 0.6, (cap, 2)), torch.float32); a = dev(rng.uniform(-1, 1, (cap, 1)), torch.float32)
r = dev(rng.normal(size=cap), torch.float32); t = dev(rng.random(cap) < 0.01, torch.uint8)
n_it = 500
idx = torch.randint(0, cap, (n_it, 64), dtype=torch.int32, device="cuda")
for _ in range(2): agent.train_on(s, a, r, t, s, idx, n_it)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); l = agent.train_on(s, a, r, t, s, idx, n_it); e1.record(); torch.cuda.synchronize()
print(sys.argv[1].split("/")[-1], "us_per_iter %%.2f  last losses %%s" %% (e0.elapsed_time(e1) / n_it * 1e3, l[-1].cpu().numpy()))
''' % ROOT
for lib in ["smartstartcontinuous_amd/libssc.so"] + sorted(__import__("glob").glob(os.path.join(ROOT, "tools/_build/libssc_ddpg[0-9]*.so"))):
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(ROOT, lib)], capture_output=True, text=True)
    print([l for l in out.stdout.splitlines() if "us_per_iter" in l] or out.stderr[-400:], flush=True)
