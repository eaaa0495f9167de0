"""Experiment driver: time ssc_rollout for several (n_envs, K) at a constant log size."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smartstartcontinuous_amd import VecEnv, RandomPolicy, TransitionChunk

def run(n, K, reps=10, log=True, env_name="MountainCarContinuous-v0", policy=None):
    env = VecEnv(env_name, n, seed=1)
    env.reset()
    chunk = TransitionChunk(env.obs_dim, K, n, env.device) if log else None
    pd = env.policy_desc(policy or RandomPolicy())
    for _ in range(3):
        env.rollout(K, out=chunk, policy_desc=pd, log=log)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.rollout(K, out=chunk, policy_desc=pd, log=log); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    med = ts[len(ts)//2]
    bps = (8*env.obs_dim+9)
    print(json.dumps(dict(env=env_name, n=n, K=K, log=log, ms_med=med, ms_min=ts[0], steps_per_s=n*K/med*1e3,
                          GBs=n*K*bps/med/1e6)), flush=True)

if __name__ == "__main__" and len(sys.argv) == 1:
    for n in (65536, 131072, 262144, 524288, 1048576):
        run(n, 1024*65536//n)
    run(65536, 1024, log=False)
    run(262144, 256, log=False)


def actor_runs():
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tests.gpu_util import actor_weights
    from smartstartcontinuous_amd import ActorPolicy
    w = {k: torch.as_tensor(v) for k, v in actor_weights(2, 64, 32, seed=1234).items()}
    for prec in ("bf16_mfma", "f32"):
        for n, K in ((65536, 256), (262144, 64)):
            print(prec, end=" ")
            run(n, K, policy=ActorPolicy(w, precision=prec))
            print(prec, "nolog", end=" ")
            run(n, K, policy=ActorPolicy(w, precision=prec), log=False)
    w3 = {k: torch.as_tensor(v) for k, v in actor_weights(3, 64, 32, seed=1234).items()}
    for n, K in ((65536, 256),):
        print("pend random", end=" "); run(n, K, env_name="Pendulum-v0")
        print("pend mfma", end=" "); run(n, K, env_name="Pendulum-v0", policy=ActorPolicy(w3, precision="bf16_mfma"))

if len(sys.argv) > 1 and sys.argv[1] == "actor":
    actor_runs()
