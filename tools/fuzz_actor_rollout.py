"""Randomised sweep of the fused actor rollout (BASELINE config 3's kernels: the shape-specialised 64-32 bf16-MFMA
policy, the generic MFMA policy and the fp32 policy) against the oracle's teacher-forced replay: random env counts, chunk
lengths, unaligned first steps, time limits inside the chunk, OU parameters / epsilon, last layer tanh or relu, observation
clip, hidden sizes.  Development tool: python tools/fuzz_actor_rollout.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import smartstartcontinuous_amd as ssc
from oracle import ssc_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)


def actor_weights(obs_dim, h1, h2, w3_scale):
    def glorot(i, o):
        lim = np.sqrt(6.0 / (i + o))
        return rng.uniform(-lim, lim, size=(i, o)).astype(np.float32)
    return dict(W1=glorot(obs_dim, h1), b1=(rng.normal(size=h1) * 0.1).astype(np.float32), W2=glorot(h1, h2),
                b2=(rng.normal(size=h2) * 0.1).astype(np.float32), W3=rng.uniform(-w3_scale, w3_scale, size=(h2, 1)).astype(np.float32),
                b3=rng.uniform(-w3_scale, w3_scale, size=1).astype(np.float32))


worst = {"f32": 0.0, "bf16_mfma": 0.0}
refused = 0
for case in range(cases):
    n = int(rng.choice([1, 31, 64, 65, 1000, 4099]))
    K = int(rng.choice([1, 3, 4, 5, 9, 16, 50]))
    h1, h2 = [(64, 32), (64, 32), (64, 64), (32, 32), (48, 24)][int(rng.integers(0, 5))]
    precision = str(rng.choice(["f32", "bf16_mfma"]))
    llt = bool(rng.integers(0, 2))
    seed, id0, t0 = int(rng.integers(0, 2**31)), int(rng.choice([0, 77, 2**33 + 1])), int(rng.integers(0, 9))
    limit = int(rng.choice([20, 999]))
    steps0 = int(rng.integers(0, limit))
    ou = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.1, 0.8)), float(rng.uniform(0.05, 0.3)), 1e-2)
    eps = float(rng.choice([0.0, 0.3, 1.0]))
    clip = float(rng.choice([0.0, 5.0, 0.4]))
    w = actor_weights(2, h1, h2, 0.5)
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed, env_id0=id0, max_episode_steps=limit)
    obs0 = env.reset().cpu().numpy()
    env.steps.fill_(steps0)
    env.t = t0
    pol = ssc.ActorPolicy({k: torch.as_tensor(v) for k, v in w.items()}, precision=precision, last_layer_tanh=llt, ou_mu=ou[0],
                          ou_sigma=ou[1], ou_theta=ou[2], ou_dt=ou[3], ou_epsilon=eps, obs_clip=clip)
    try:
        chunk = env.rollout(K, pol)
    except ssc._ffi.SscError as e:       # the fused fp32 policy exists for 64-32 only: a clean refusal is the contract
        assert precision == "f32" and (h1, h2) != (64, 32) and "64-32" in str(e), (case, e)
        refused += 1
        print("case %2d nets %d-%d f32: refused (SSC_EUNSUPPORTED)" % (case, h1, h2), flush=True)
        continue
    torch.cuda.synchronize()
    log = dict(obs=chunk.obs.cpu().numpy(), act=chunk.act.cpu().numpy(), rew=chunk.rew.cpu().numpy(), done=chunk.done.cpu().numpy(),
               obs2=chunk.obs2.cpu().numpy())
    opol = O.OracleDDPGPolicy(w, seed, id0, n, ou=ou, epsilon=eps, last_layer_tanh=llt, bf16=False, obs_clip=clip or None)
    res = O.replay_rollout("mc", log, seed, id0, t0, limit, obs0, np.full(n, steps0), opol)
    tol = 2e-5 if precision == "f32" else 4e-2
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0 and res["reset_max_err"] == 0, (case, res)
    assert res["max_dact"] <= tol and res["max_dobs2"][0] <= 2.4e-7 and res["max_dobs2"][1] <= 1e-8 and res["max_drew_rel"] <= 1e-6, (case, precision, res)
    if eps > 0:
        # (the oracle policy clears the OU state of a finished episode at its NEXT call, the kernel right away)
        x_ref = np.where(log["done"][-1].astype(bool), 0.0, opol.x)
        assert np.max(np.abs(env.ou_x.cpu().numpy() - x_ref)) < 2e-5, (case, "ou state", n, K, steps0, limit, t0)
    worst[precision] = max(worst[precision], res["max_dact"])
    print("case %2d n %4d K %2d nets %d-%d %-9s tanh %d eps %.1f clip %.1f limit %3d: max |d action| %.1e" %
          (case, n, K, h1, h2, precision, llt, eps, clip, limit, res["max_dact"]), flush=True)
print("actor rollout: %d random configurations ok (%d refused: fp32 policy with nets other than 64-32), worst action deviation fp32 %.1e / bf16 %.1e"
      % (cases, refused, worst["f32"], worst["bf16_mfma"]))
