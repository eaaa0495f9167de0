"""Randomised sweep of the fused random-policy rollout (BASELINE config 2's kernel) against the oracle's teacher-forced
replay: ragged env counts, chunk lengths on and off the 4-step Philox grouping, unaligned first steps, env-id offsets,
time limits that fall inside the chunk, power scalars; statistics and episode records are checked against the log.
Development tool: python tools/fuzz_rollout.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import smartstartcontinuous_amd as ssc
from oracle import ssc_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
for case in range(cases):
    n = int(rng.choice([1, 63, 64, 65, 255, 1000, 4099, 20000]))
    K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 33, 64, 129]))
    seed, id0 = int(rng.integers(0, 2**31)), int(rng.choice([0, 1, 10**6, 2**33 + 5]))
    t0 = int(rng.integers(0, 50))
    max_steps = int(rng.choice([0, 5, 50, 999]))
    steps0 = int(rng.integers(0, max(1, max_steps))) if max_steps else int(rng.integers(0, 2000))
    ps = float(rng.choice([1.0, 1.3, 2.0]))
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed, env_id0=id0, max_episode_steps=max_steps, power_scalar=ps)   # 0 = no time limit
    env.reset()
    env.steps.fill_(steps0)
    env.t = t0
    pos0, vel0 = env.s0.cpu().numpy(), env.s1.cpu().numpy()
    ring = ssc.EpisodeRing(1 << 16, "cuda")
    chunk = env.rollout(K, ssc.RandomPolicy(), ring=ring)
    torch.cuda.synchronize()
    log = dict(s_pos=chunk.obs[0].cpu().numpy(), s_vel=chunk.obs[1].cpu().numpy(), act=chunk.act.cpu().numpy(),
               rew=chunk.rew.cpu().numpy(), done=chunk.done.cpu().numpy(), s2_pos=chunk.obs2[0].cpu().numpy(),
               s2_vel=chunk.obs2[1].cpu().numpy())
    res = O.mc_replay_random_rollout(log, seed, id0, t0, O.mc_power(ps), max_steps or (1 << 60), pos0, vel0, np.full(n, steps0, np.int64))
    bad = {k: res[k] for k in ("start_mismatch", "act_mismatch", "done_mismatch", "continuity_mismatch", "reset_mismatch") if res[k]}
    assert not bad and res["max_dpos"] <= 2.4e-7 and res["max_dvel"] <= 1e-8 and res["max_drew"] <= 1e-4, (case, n, K, res)
    assert np.array_equal(env.steps.cpu().numpy(), res["final_elapsed"]), (case, "elapsed")
    stats = env.stats.cpu().numpy()
    n_done = int(log["done"].sum())
    assert stats[2] == n * K and stats[3] == n_done and abs(stats[0] - log["rew"].astype(np.float64).sum()) < 1e-2 + 1e-6 * n * K
    (eid, elen, eret), dropped = ring.drain()
    assert dropped == 0 and len(eid) == n_done, (case, len(eid), n_done)
    print("case %2d n %5d K %3d t0 %2d steps0 %4d limit %3d power x%.1f id0 %d: %d episodes ended, ok" % (case, n, K, t0, steps0, max_steps, ps, id0, n_done),
          flush=True)
print("random-policy rollout: %d random configurations ok" % cases)

# Pendulum-v0 (the data-collection env of NND_MB_agent): same sweep, generic replay
for case in range(cases // 2):
    n = int(rng.choice([1, 63, 64, 65, 777, 4099]))
    K = int(rng.choice([1, 3, 4, 5, 8, 40, 129]))
    seed, id0, t0 = int(rng.integers(0, 2**31)), int(rng.choice([0, 5, 2**33 + 5])), int(rng.integers(0, 50))
    limit = int(rng.choice([7, 200]))
    steps0 = int(rng.integers(0, limit))
    env = ssc.VecEnv("Pendulum-v0", n, seed=seed, env_id0=id0, max_episode_steps=limit)
    obs0 = env.reset().cpu().numpy()
    env.steps.fill_(steps0)
    env.t = t0
    chunk = env.rollout(K, ssc.RandomPolicy())
    torch.cuda.synchronize()
    log = dict(obs=chunk.obs.cpu().numpy(), act=chunk.act.cpu().numpy(), rew=chunk.rew.cpu().numpy(), done=chunk.done.cpu().numpy(),
               obs2=chunk.obs2.cpu().numpy())
    res = O.replay_rollout("pend", log, seed, id0, t0, limit, obs0, np.full(n, steps0), O.OracleRandomPolicy(seed, id0, n, -2.0, 2.0))
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0 and res["max_dact"] == 0, (case, res)
    assert (res["max_dobs2"] <= [3e-6, 3e-6, 3e-6]).all() and res["max_drew_rel"] <= 2e-5 and res["reset_max_err"] <= 2e-6, (case, res)
    print("pend case %2d n %4d K %3d t0 %2d steps0 %3d limit %3d id0 %d: ok" % (case, n, K, t0, steps0, limit, id0), flush=True)
print("Pendulum random-policy rollout: %d random configurations ok" % (cases // 2))
