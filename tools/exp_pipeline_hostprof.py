"""Host-side profile of rl_train_vec_ddpg (where does a chunk's millisecond go when the GPU work is 0.5 ms?)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
env = ssc.VecEnv("MountainCarContinuous-v0", 4096, seed=1); env.reset()
agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=64, num_train_iterations=50,
                             actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=3)
ssc.rl_train_vec_ddpg(env, agent, num_chunks=3, chunk_steps=64, replay_capacity=1 << 20, replay_last_steps=16)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
ssc.rl_train_vec_ddpg(env, agent, num_chunks=50, chunk_steps=64, replay_capacity=1 << 20, replay_last_steps=16)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
