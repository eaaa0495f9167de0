"""Fused rollout with the wide actors (128-64 register-resident, 200-100 LDS-staged) at the BASELINE env count:
us per 65 536 x 256 env-steps.  SSC_LIB_PATH selects a variant library."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import init_actor_weights
n, K = 65536, 256
for h1, h2 in ((200, 100), (128, 64)):
    w = init_actor_weights(2, h1, h2, 1, torch.Generator().manual_seed(1234))
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=1234)
    env.reset()
    chunk = ssc.TransitionChunk(2, K, n, env.device)
    pd = env.policy_desc(ssc.ActorPolicy(w, precision="bf16_mfma", ou_mu=0.4, ou_sigma=0.6, ou_theta=0.15, obs_clip=5.0))
    for _ in range(20):
        env.rollout(K, out=chunk, policy_desc=pd)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); env.rollout(K, out=chunk, policy_desc=pd); b.record()
    torch.cuda.synchronize()
    d = sorted(a.elapsed_time(b) for a, b in evs)
    med = d[len(d) // 2]
    flop = 2.0 * (2 * h1 + h1 * h2 + h2)
    print(json.dumps({"actor": "%d-%d" % (h1, h2), "lib": os.environ.get("SSC_LIB_PATH", "committed"), "ms_per_launch_median": med, "min": d[0],
                      "env_steps_per_s": n * K / (med * 1e-3), "tflops": flop * n * K / (med * 1e-3) / 1e12}), flush=True)
