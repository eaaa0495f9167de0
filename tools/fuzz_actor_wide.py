"""Randomised sweep of the fused actor rollout for the WIDER networks of the reference's grid (register-resident MFMA policy
up to 128-64, LDS-staged policy up to 224-128) on both envs, against the oracle's teacher-forced replay: random env counts,
chunk lengths, unaligned first steps, time limits inside the chunk, OU parameters / epsilon, last layer tanh or relu,
observation clip, hidden sizes on and off the 32-unit tiles.  Development tool: python tools/fuzz_actor_wide.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import smartstartcontinuous_amd as ssc
from oracle import ssc_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)


def actor_weights(obs_dim, h1, h2, w3_scale):
    def glorot(i, o):
        lim = np.sqrt(6.0 / (i + o))
        return rng.uniform(-lim, lim, size=(i, o)).astype(np.float32)
    return dict(W1=glorot(obs_dim, h1), b1=(rng.normal(size=h1) * 0.1).astype(np.float32), W2=glorot(h1, h2),
                b2=(rng.normal(size=h2) * 0.1).astype(np.float32), W3=rng.uniform(-w3_scale, w3_scale, size=(h2, 1)).astype(np.float32),
                b3=rng.uniform(-w3_scale, w3_scale, size=1).astype(np.float32))


worst = 0.0
for case in range(cases):
    pend = bool(rng.integers(0, 2))
    n = int(rng.choice([1, 63, 64, 129, 1000, 2051]))
    K = int(rng.choice([1, 2, 5, 8, 17]))
    h1, h2 = [(128, 64), (200, 100), (224, 128), (160, 96), (100, 100), (129, 33), (65, 64)][int(rng.integers(0, 7))]
    llt = bool(rng.integers(0, 2))
    seed, id0, t0 = int(rng.integers(0, 2**31)), int(rng.choice([0, 5, 2**33 + 3])), int(rng.integers(0, 9))
    limit = int(rng.choice([15, 200 if pend else 999]))
    steps0 = int(rng.integers(0, limit))
    ou = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.1, 0.8)), float(rng.uniform(0.05, 0.3)), 1e-2)
    eps = float(rng.choice([0.0, 0.3, 1.0]))
    clip = float(rng.choice([0.0, 5.0, 0.8]))
    obs_dim = 3 if pend else 2
    w = actor_weights(obs_dim, h1, h2, 0.4)
    env = ssc.VecEnv("Pendulum-v0" if pend else "MountainCarContinuous-v0", n, seed=seed, env_id0=id0, max_episode_steps=limit)
    obs0 = env.reset().cpu().numpy()
    env.steps.fill_(steps0)
    env.t = t0
    pol = ssc.ActorPolicy({k: torch.as_tensor(v) for k, v in w.items()}, precision="bf16_mfma", last_layer_tanh=llt, ou_mu=ou[0],
                          ou_sigma=ou[1], ou_theta=ou[2], ou_dt=ou[3], ou_epsilon=eps, obs_clip=clip)
    chunk = env.rollout(K, pol)
    torch.cuda.synchronize()
    log = dict(obs=chunk.obs.cpu().numpy(), act=chunk.act.cpu().numpy(), rew=chunk.rew.cpu().numpy(), done=chunk.done.cpu().numpy(),
               obs2=chunk.obs2.cpu().numpy())
    low, high = (-2.0, 2.0) if pend else (-1.0, 1.0)
    opol = O.OracleDDPGPolicy(w, seed, id0, n, ou=ou, epsilon=eps, low=low, high=high, last_layer_tanh=llt, bf16=False, obs_clip=clip or None)
    res = O.replay_rollout("pend" if pend else "mc", log, seed, id0, t0, limit, obs0, np.full(n, steps0), opol)
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0, (case, res)
    assert res["reset_max_err"] <= (2e-6 if pend else 0.0), (case, res)     # Pendulum observations are cos / sin of the reset angle
    assert res["max_dact"] <= 4e-2 * (high - low) / 2, (case, h1, h2, res)
    if eps > 0:
        x_ref = np.where(log["done"][-1].astype(bool), 0.0, opol.x)
        assert np.max(np.abs(env.ou_x.cpu().numpy() - x_ref)) < 2e-5, (case, "ou state")
    worst = max(worst, res["max_dact"] / ((high - low) / 2))
    print("case %2d %-9s n %4d K %2d nets %d-%d tanh %d eps %.1f clip %.1f limit %3d: max |d action| %.1e" %
          (case, "Pendulum" if pend else "MountainCar", n, K, h1, h2, llt, eps, clip, limit, res["max_dact"]), flush=True)
print("wide actor rollout: %d random configurations ok, worst action deviation %.1e of the action range" % (cases, worst))
