"""One rank of the multi-rank GPU tests (tests/test_gpu_multirank.py): started as a FRESH child process per rank by
tests/conftest.py before the pytest process touches the GPU; gloo process group (the packed payloads travel through
pinned host memory -- ``TransitionGather(host_staging)``), every rank on GPU 0.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment;  argv[1] = output directory.

Runs ``rl_train_sharded_ddpg`` over ONE global id space of N_TOTAL envs cut into ``shard_range`` shards -- synchronous
and pipelined -- and writes, per rank, everything the test compares with the world-1 run of the same id space: the full
transition log of every chunk, the final env state, the synchronised array [actor | epsilon], and on the learner the
replay ring, the critic and the losses."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_TOTAL, CHUNK_STEPS, G_STEPS, NUM_CHUNKS, MAX_EPISODE_STEPS = 769, 48, 8, 6, 60


def make(ssc, lo, hi):
    from smartstartcontinuous_amd.agents import DDPG_Baselines_agent
    env = ssc.VecEnv("MountainCarContinuous-v0", hi - lo, seed=11, max_episode_steps=MAX_EPISODE_STEPS, env_id0=lo)
    one = ssc.SingleEnvView(ssc.VecEnv("MountainCarContinuous-v0", 1, seed=11))
    # a decay the six chunks can see: 288 steps = 4 generations of 60-step episodes
    agent = DDPG_Baselines_agent(one, None, batch_size=64, num_train_iterations=5, actor_h1=64, actor_h2=32, critic_h1=64,
                                 critic_h2=32, lastLayerTanh=True, seed=4, ou_epsilon_decay_factor=0.8)
    return env, agent


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import smartstartcontinuous_amd as ssc
    from smartstartcontinuous_amd.sharding import rl_train_sharded_ddpg, shard_range
    lo, hi = shard_range(N_TOTAL, world, rank)
    res = {"lo": lo, "hi": hi}
    for mode in ("sync", "pipelined", "pipelined_again"):
        env, agent = make(ssc, lo, hi)
        logs = []

        def keep(i, chunk, env_):
            logs.append([x.clone() for x in (chunk.obs, chunk.act, chunk.rew, chunk.done, chunk.obs2)])
        summary, losses, replay = rl_train_sharded_ddpg(
            env, agent, num_chunks=NUM_CHUNKS, chunk_steps=CHUNK_STEPS, rank=rank, world=world, gather_steps=G_STEPS,
            replay_capacity=1 << 16, seed=3, pipelined=mode != "sync", on_chunk=keep, drain_every=2)
        torch.cuda.synchronize()
        for c, name in enumerate(("obs", "act", "rew", "done", "obs2")):
            res[f"{mode}_log_{name}"] = torch.stack([l[c] for l in logs]).cpu().numpy()
        res[f"{mode}_s0"], res[f"{mode}_s1"] = env.s0.cpu().numpy(), env.s1.cpu().numpy()
        res[f"{mode}_ou_x"], res[f"{mode}_steps"] = env.ou_x.cpu().numpy(), env.steps.cpu().numpy()
        res[f"{mode}_actor_sync"] = agent.actor_sync.cpu().numpy()
        res[f"{mode}_host_epsilon"] = np.float64(agent.decaying_ou_action_noise.epsilon)
        res[f"{mode}_episodes"] = np.array(sorted(summary.episodes), np.float64).reshape(-1, 2)
        res[f"{mode}_stats"] = env.stats.cpu().numpy()
        if replay is not None:
            res[f"{mode}_replay_count"] = np.int64(replay.count)
            for name in ("s", "a", "r", "t", "s2"):
                res[f"{mode}_replay_{name}"] = getattr(replay, name).cpu().numpy()
            res[f"{mode}_critic"] = agent.critic_flat.cpu().numpy()
            res[f"{mode}_losses"] = torch.stack(losses).cpu().numpy()
    np.savez(os.path.join(out_dir, f"world{world}_rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
