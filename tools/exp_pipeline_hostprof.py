"""Host-side profile of rl_train_vec_ddpg (where does a chunk's millisecond go when the GPU work is 0.5 ms?)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
N = int(os.environ.get("SSC_PIPE_ENVS", "4096")); K = int(os.environ.get("SSC_PIPE_STEPS", "64"))
OVERLAP = os.environ.get("SSC_PIPE_OVERLAP", "0") == "1"
env = ssc.VecEnv("MountainCarContinuous-v0", N, seed=1); env.reset()
agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=64, num_train_iterations=50,
                             actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32, lastLayerTanh=True, seed=3)
ssc.rl_train_vec_ddpg(env, agent, num_chunks=3, chunk_steps=K, replay_capacity=1 << 20, replay_last_steps=16, overlap=OVERLAP)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
ssc.rl_train_vec_ddpg(env, agent, num_chunks=50, chunk_steps=K, replay_capacity=1 << 20, replay_last_steps=16, overlap=OVERLAP)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
