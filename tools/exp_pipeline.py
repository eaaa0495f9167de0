"""End-to-end actor-learner loop in HBM (rl_train_vec_ddpg): fused rollout under the current actor -> device replay
ring -> DDPG iterations.  Prints env-steps/s and learner iterations/s for a few chunk / iteration mixes."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent

# (envs, chunk steps, learner iterations per chunk, overlap, batch, actor/critic hidden sizes); batches != 64 and layers wider than
# 64 run on the multi-workgroup learner (ddpg_train_wide.hip): batch / 16 workgroups per iteration
CASES = [(65536, 256, 50, False, 64, (64, 32)), (65536, 256, 50, True, 64, (64, 32)), (65536, 256, 500, False, 64, (64, 32)),
         (65536, 256, 500, True, 64, (64, 32)), (4096, 64, 50, False, 64, (64, 32)), (4096, 64, 50, True, 64, (64, 32)),
         (256, 40, 50, False, 64, (64, 32)),
         (65536, 256, 10, False, 1024, (64, 32)), (65536, 256, 10, False, 4096, (64, 32)), (65536, 256, 50, False, 4096, (64, 32)),
         (65536, 256, 10, False, 1024, (200, 100)), (65536, 256, 10, False, 4096, (200, 100)),
         (65536, 256, 10, True, 1024, (64, 32)), (65536, 256, 25, True, 1024, (64, 32)), (65536, 256, 25, False, 1024, (64, 32))]
if len(sys.argv) > 1 and sys.argv[1] == "wide":
    CASES = CASES[7:]
if len(sys.argv) > 1 and sys.argv[1] == "overlap":
    CASES = [CASES[7]] + CASES[12:]
for n_envs, chunk, iters, overlap, batch, (h1, h2) in CASES:
    env = ssc.VecEnv("MountainCarContinuous-v0", n_envs, seed=1)
    env.reset()
    agent = DDPG_Baselines_agent(ssc.make("MountainCarContinuous-v0"), None, batch_size=batch, num_train_iterations=iters,
                                 actor_h1=h1, actor_h2=h2, critic_h1=h1, critic_h2=h2, lastLayerTanh=True, seed=3)
    ssc.rl_train_vec_ddpg(env, agent, num_chunks=2, chunk_steps=chunk, replay_capacity=1 << 20, replay_last_steps=16,
                          overlap=overlap)
    torch.cuda.synchronize()
    n_chunks = 200 if iters <= 50 else 40     # the call's set-up (ring, replay, chunk allocation) is ~10 ms: amortise it
    t0 = time.perf_counter()
    summary, losses, replay = ssc.rl_train_vec_ddpg(env, agent, num_chunks=n_chunks, chunk_steps=chunk,
                                                    replay_capacity=1 << 20, replay_last_steps=16, overlap=overlap)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(n_envs=n_envs, chunk_steps=chunk, overlap=overlap, train_iters_per_chunk=iters, batch=batch, nets="%d-%d" % (h1, h2),
                          samples_trained_per_env_step=iters * batch / float(n_envs * chunk), ms_per_chunk=dt / n_chunks * 1e3,
                          env_steps_per_s=n_envs * chunk * n_chunks / dt, learner_iters_per_s=iters * n_chunks / dt,
                          episodes=len(summary.episodes))), flush=True)
