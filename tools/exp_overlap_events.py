"""Chunk period of rl_train_vec_ddpg with and without overlap, from GPU events recorded behind every learner launch
(no profiler, no host timing): python tools/exp_overlap_events.py [n_envs chunk_steps]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for overlap in (False, True, False, True):
    env = ssc.VecEnv("MountainCarContinuous-v0", n_envs, seed=1); env.reset()
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=64, num_train_iterations=50,
                                 actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=3)
    evs = []
    orig = agent.train_from
    def train_from(replay, iters=None, _orig=orig):
        l = _orig(replay, iters)
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        return l
    agent.train_from = train_from
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ssc.rl_train_vec_ddpg(env, agent, num_chunks=120, chunk_steps=chunk, replay_capacity=1 << 20, replay_last_steps=16, overlap=overlap)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    d = sorted(a.elapsed_time(b) for a, b in zip(evs[20:-1], evs[21:]))
    print(json.dumps(dict(n_envs=n_envs, chunk_steps=chunk, overlap=overlap, chunk_period_ms_median=d[len(d) // 2], p10=d[len(d) // 10],
                          p90=d[9 * len(d) // 10], max=d[-1], mean=sum(d) / len(d), wall_ms_per_chunk=wall / 120 * 1e3,
                          first_to_last_train_ms_per_chunk=evs[0].elapsed_time(evs[-1]) / (len(evs) - 1))), flush=True)
