#!/usr/bin/env python3
"""rl_train_vec_ddpg at 65 536 envs x 256-step chunks, 10 x batch 1024 per chunk (for rocprofv3 --kernel-trace --stats): prints
ms per chunk; the kernel trace says how much of a chunk is kernels and how much is launch boundaries."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
env = ssc.VecEnv("MountainCarContinuous-v0", 65536, seed=1)
env.reset()
agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=1024, num_train_iterations=10, actor_h1=64, actor_h2=32,
                             critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=3)
ssc.rl_train_vec_ddpg(env, agent, num_chunks=5, chunk_steps=256, replay_capacity=1 << 20, replay_last_steps=16)
torch.cuda.synchronize()
t0 = time.perf_counter()
ssc.rl_train_vec_ddpg(env, agent, num_chunks=n_chunks, chunk_steps=256, replay_capacity=1 << 20, replay_last_steps=16)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"ms_per_chunk": dt / n_chunks * 1e3, "env_steps_per_s": 65536 * 256 * n_chunks / dt, "chunks": n_chunks}))
