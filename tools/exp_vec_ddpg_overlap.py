#!/usr/bin/env python3
"""rl_train_vec_ddpg at 65 536 envs x 256-step chunks, 10 x batch 1024: overlap=False against overlap=True (rollout of chunk i + 1 on a
second stream while the learner works on chunk i), alternating on one box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent

def run(overlap, batch=1024, iters=10, n_chunks=200):
    env = ssc.VecEnv("MountainCarContinuous-v0", 65536, seed=1)
    env.reset()
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=batch, num_train_iterations=iters, actor_h1=64, actor_h2=32,
                                 critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=3)
    kw = dict(chunk_steps=256, replay_capacity=1 << 20, replay_last_steps=16, overlap=overlap)
    ssc.rl_train_vec_ddpg(env, agent, num_chunks=5, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ssc.rl_train_vec_ddpg(env, agent, num_chunks=n_chunks, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / n_chunks * 1e3

for rep in range(2):
    for ov in (False, True):
        print(json.dumps({"overlap": ov, "batch": 1024, "iters": 10, "ms_per_chunk": round(run(ov), 4)}), flush=True)
for ov in (False, True):
    print(json.dumps({"overlap": ov, "batch": 1024, "iters": 25, "ms_per_chunk": round(run(ov, iters=25), 4)}), flush=True)
