"""Stage times of the device SmartStart selection (smartstart.device_smart_start_path) on a 100 000-record device ring."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd import smartstart as SS
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer
n_envs, K, cap, n_ss = 512, 250, 100000, 2000
env = ssc.VecEnv("MountainCarContinuous-v0", n_envs, seed=3, max_episode_steps=120); env.reset()
agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, actor_h1=64, actor_h2=32, critic_h1=64, critic_h2=32,
                             lastLayerTanh=True, seed=1, training=False)
replay = DeviceReplayBuffer(cap, 2, 1, env.device, seed=5, track_episodes=True, n_envs=n_envs, max_path_len=130)
for _ in range(2):
    replay.append_chunk(env.rollout(K, ssc.RandomPolicy()), reward_scale=1.0)
torch.cuda.synchronize()
radii = np.array([0.05, 0.005])
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
out = {}
out["whole_ms"], res = timed(lambda: SS.device_smart_start_path(replay, agent, radii, n_ss))
out["indices_ms"], idx = timed(lambda: replay.get_possible_smart_start_indices(n_ss))
alls = replay.get_all_states()
out["scott_ms"], (wh, norm) = timed(lambda: SS.kde_scott_bandwidth(alls))
cand = replay.s2[replay.physical(idx)]
out["value_ms"], vals = timed(lambda: agent.state_value_device(cand))
out["kde_ms"], pdf = timed(lambda: SS.kde_evaluate(alls, cand, wh, norm))
out["ucb_ms"], (_, best) = timed(lambda: SS.ucb_argmax(vals, pdf, len(replay), 1.0, 1.0, 2.0))
out["path_ms"], _ = timed(lambda: replay.get_episodic_path_to_buffer_index(idx[best.long()]))
out["len_replay"], out["n_candidates"] = len(replay), int(idx.numel())
print(json.dumps(out))
