"""Time of one learner iteration (ssc_ddpg_train_ws) for the 64-32 networks by batch size: the 64-row-tile straight-line
kernel + apply pass against the 16-row-tile kernel (SSC_DDPG_WIDE=1) and, at batch 64, the single-workgroup kernel."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent

def run(B, wide, obs_env="MountainCarContinuous-v0", iters=50, reps=20):
    if wide: os.environ["SSC_DDPG_WIDE"] = "1"
    else: os.environ.pop("SSC_DDPG_WIDE", None)
    env = ssc.make(obs_env)
    agent = DDPG_Baselines_agent(env, None, batch_size=B, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=1, training=False)
    od = env.observation_space.shape[0]
    cap = 1 << 16
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    s = torch.randn((cap, od), device="cuda", generator=g); a = torch.rand((cap, 1), device="cuda", generator=g) * 2 - 1
    r = torch.randn(cap, device="cuda", generator=g); t = (torch.rand(cap, device="cuda", generator=g) < 0.05).to(torch.uint8)
    s2 = s + 0.01
    idx = torch.randint(0, cap, (iters, B), device="cuda", generator=g, dtype=torch.int32)
    for _ in range(3): agent.train_on(s, a, r, t, s2, idx, iters)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): agent.train_on(s, a, r, t, s2, idx, iters)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * iters)

out = []
for B in (64, 128, 256, 1024, 4096):
    row = {"batch": B, "us_per_iter": round(run(B, False), 2), "us_per_iter_16_row_tiles": round(run(B, True), 2)}
    print(json.dumps(row), flush=True); out.append(row)
