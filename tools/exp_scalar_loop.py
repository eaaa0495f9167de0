"""The reference's own execution model on this engine: scalar rlTrain (one env, one action per call, one DDPG train
iteration per env step) -- host-side profile of where a step's time goes."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
env = ssc.make("MountainCarContinuous-v0")
H1, H2 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 32)
agent = DDPG_Baselines_agent(env, None, actor_h1=H1, actor_h2=H2, critic_h1=H1, critic_h2=H2, lastLayerTanh=True, seed=3,
                             batch_size=64, num_train_iterations=1)
ssc.rlTrain(agent, env, num_episodes=1, max_steps=300, print_steps=False, print_results=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
summary = ssc.rlTrain(agent, env, num_episodes=2, max_steps=500, print_steps=False, print_results=False)
pr.disable()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = sum(e[0] if isinstance(e, (tuple, list)) else len(e) for e in summary.episodes)
print("%d-%d" % (H1, H2), "scalar rlTrain + DDPG (train every step): %d steps in %.2f s = %.0f steps/s (%.2f ms per step)" % (steps, dt, steps / dt, dt / steps * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
