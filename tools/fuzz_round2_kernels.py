"""Randomised differential check of the kernels touched in round 2 against the fp64 oracle (run on the GPU box; not
part of the test suite -- the fixed cases in tests/ are): the DDPG learner on random layer sizes (step interpreter) and
on the shipped shape (specialised kernel), the KDE for every dimension, the device episode index + smart-start sampling.
usage: python tools/fuzz_round2_kernels.py [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ssc_oracle as O  # noqa: E402
import smartstartcontinuous_amd as ssc  # noqa: E402
from smartstartcontinuous_amd import smartstart as SS  # noqa: E402
from smartstartcontinuous_amd import spaces  # noqa: E402
from smartstartcontinuous_amd.agents import DDPG_Baselines_agent  # noqa: E402
from smartstartcontinuous_amd.replay_buffer import DeviceReplayBuffer  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
dev = lambda x, dt: torch.as_tensor(x, dtype=dt, device="cuda").contiguous()


class BoxEnv:
    def __init__(self, d):
        self.observation_space = spaces.Box(low=-np.ones(d, np.float32), high=np.ones(d, np.float32))
        self.action_space = spaces.Box(low=np.array([-1.0], np.float32), high=np.array([1.0], np.float32))


# ---- DDPG learner ---------------------------------------------------------------------------------------------------
worst, refused = 0.0, 0
for trial in range(24):
    d = int(rng.integers(1, 9))
    shipped = trial % 4 == 0
    h1, h2 = (64, 32) if shipped else (int(rng.integers(1, 65)), int(rng.integers(1, 65)))
    if shipped:
        d = int(rng.choice([2, 3]))
    llt = bool(rng.integers(0, 2))
    agent = DDPG_Baselines_agent(BoxEnv(d), None, actor_h1=h1, actor_h2=h2, critic_h1=h1, critic_h2=h2, lastLayerTanh=llt,
                                 actor_lr=1e-3, critic_lr=1e-3, gamma=0.99, tau=0.01, batch_size=64, seed=int(rng.integers(1 << 30)),
                                 training=False)
    aw = {k: v.cpu().numpy() + (0.05 * rng.normal(size=tuple(v.shape))).astype(np.float32) for k, v in agent.weights.items()}
    cw = {k: v.cpu().numpy() + (0.05 * rng.normal(size=tuple(v.shape))).astype(np.float32) for k, v in agent.critic_weights.items()}
    agent.set_weights(aw); agent.set_critic_weights(cw)
    cap, n_it = 500, int(rng.integers(1, 6))
    s = rng.uniform(-2, 2, (cap, d)).astype(np.float32); a = rng.uniform(-1, 1, (cap, 1)).astype(np.float32)
    r = (rng.normal(size=cap) * 0.5).astype(np.float32); t = rng.random(cap) < 0.2
    s2 = (s + rng.normal(size=(cap, d)) * 0.05).astype(np.float32)
    idx = np.stack([rng.permutation(cap)[:64] for _ in range(n_it)]).astype(np.int32)
    o_a = {k: v.astype(np.float64) for k, v in aw.items()}; o_c = {k: v.astype(np.float64) for k, v in cw.items()}
    o_ta = O.unflatten_params(agent.target_actor_flat.cpu().numpy().astype(np.float64), o_a)
    o_tc = O.unflatten_params(agent.target_critic_flat.cpu().numpy().astype(np.float64), o_c)
    na, nc = agent.actor_flat.numel(), agent.critic_flat.numel()
    adam = dict(m_actor=np.zeros(na), v_actor=np.zeros(na), t_actor=0, m_critic=np.zeros(nc), v_critic=np.zeros(nc), t_critic=0)
    for it in range(n_it):
        bi = idx[it]
        o_a, o_c, o_ta, o_tc, adam, cl, al = O.ddpg_train_step(o_a, o_c, o_ta, o_tc, adam, (s[bi], a[bi], r[bi], t[bi], s2[bi]), gamma=0.99,
                                                               tau=0.01, actor_lr=1e-3, critic_lr=1e-3, last_layer_tanh=llt, obs_clip=5.0)
    try:
        agent.train_on(dev(s, torch.float32), dev(a, torch.float32), dev(r, torch.float32), dev(t, torch.uint8), dev(s2, torch.float32),
                       dev(idx, torch.int32), n_it)
    except ssc._ffi.SscError as e:      # layer sizes whose activations + parameters exceed the LDS are refused, loudly
        assert "LDS" in str(e), e
        refused += 1
        continue
    torch.cuda.synchronize()
    dv = max(np.max(np.abs(agent.actor_flat.cpu().numpy() - O.flatten_params(o_a))), np.max(np.abs(agent.critic_flat.cpu().numpy() - O.flatten_params(o_c))),
             np.max(np.abs(agent.target_critic_flat.cpu().numpy() - O.flatten_params(o_tc))))
    worst = max(worst, dv)
    assert dv <= 1e-5, (d, h1, h2, llt, n_it, dv)
print("DDPG learner: 24 random shapes ok (%d refused: do not fit the LDS), worst parameter deviation %.2e" % (refused, worst))

# ---- KDE ------------------------------------------------------------------------------------------------------------
worst = 0.0
for trial in range(20):
    d, n, m = int(rng.integers(1, 7)), int(rng.integers(50, 30000)), int(rng.integers(1, 300))
    data = (np.cumsum(rng.normal(size=(n, d)) * 0.01, axis=0) % 1.0).astype(np.float32)
    pts = data[rng.integers(0, n, m)]
    wh, norm = SS.kde_scott_bandwidth(dev(data, torch.float32))
    pdf = SS.kde_evaluate(dev(data, torch.float32), dev(pts, torch.float32), wh, norm).cpu().numpy()
    _c, wh_ref, norm_ref = O.kde_scott(data)
    ref = O.kde_evaluate(data, pts, wh_ref, norm_ref)
    dv = float(np.max(np.abs(pdf / ref - 1)))
    worst = max(worst, dv)
    assert dv <= 5e-4, (d, n, m, dv)
print("KDE: 20 random (d, n, m) ok, worst relative deviation %.2e" % worst)

# ---- device episode index + smart-start sampling ---------------------------------------------------------------------
for trial in range(8):
    n_envs, K, cap = int(rng.integers(1, 200)), int(rng.integers(3, 60)), int(rng.integers(200, 5000))
    env = ssc.VecEnv("MountainCarContinuous-v0", n_envs, seed=int(rng.integers(1 << 30)), max_episode_steps=int(rng.integers(4, 30)))
    env.reset()
    replay = DeviceReplayBuffer(cap, 2, 1, env.device, seed=int(rng.integers(1 << 30)), track_episodes=True, n_envs=n_envs, max_path_len=40)
    done_all = []
    ring_steps = np.zeros(cap, np.int64)
    run, count = None, 0
    for c in range(int(rng.integers(1, 5))):
        chunk = env.rollout(K, ssc.RandomPolicy())
        replay.append_chunk(chunk, reward_scale=1.0)
        st_, run = O.replay_episode_steps(chunk.done.cpu().numpy(), run)
        flat = st_.reshape(-1)                                            # record order: step-major, env-minor
        for j, v in enumerate(flat[-cap:] if flat.size > cap else flat):
            ring_steps[(count + (flat.size - min(flat.size, cap)) + j) % cap] = v
        count += flat.size
    assert np.array_equal(replay.ep_steps.cpu().numpy()[: min(count, cap)] if count <= cap else replay.ep_steps.cpu().numpy(),
                          ring_steps[: min(count, cap)] if count <= cap else ring_steps), trial
    valid = O.smart_start_valid(ring_steps, cap, count, n_envs)
    n_ss = int(rng.integers(1, 200))
    q = replay._queries
    got = replay.get_possible_smart_start_indices(n_ss)
    ref = O.smart_start_indices(valid, n_ss, replay.seed ^ 0x5353, q)
    ref = ref[ref >= 0]
    if got is None:
        assert ref.size == 0, trial
        continue
    assert np.array_equal(got.cpu().numpy(), ref), trial
print("device episode index + smart-start sampling: 8 random rings ok")
