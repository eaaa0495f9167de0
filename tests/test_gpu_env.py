"""GPU parity tests of the env kernels through the C ABI (BASELINE configs 1, 2, 5-by-sharding).
Everything is compared with oracle/ssc_oracle.py on the same seeded inputs."""
import ctypes

import numpy as np
import pytest

from oracle import ssc_oracle as O
from tests.gpu_util import assert_replay_clean, mc_log

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL_VEL, TOL_POS = 1e-8, 2.4e-7


@pytest.fixture(scope="module")
def ssc():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    import smartstartcontinuous_amd as pkg
    pkg._ffi.lib()
    return pkg


def _dev(a):
    return torch.as_tensor(a, device="cuda")


def _mc_step_gpu(ssc, pos, vel, act, params=None, steps=None):
    ffi = ssc._ffi
    p = params or ffi.default_params(ffi.SSC_ENV_MOUNTAINCAR, 1.0, 999)
    n = pos.size
    dp, dv, da = _dev(pos.copy()), _dev(vel.copy()), _dev(act)
    rew = torch.empty(n, dtype=torch.float32, device="cuda")
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    ds = _dev(steps.copy()) if steps is not None else None
    ffi.check(ffi.lib().ssc_mc_step(ctypes.byref(p), n, ffi.ptr(dp), ffi.ptr(dv), ffi.ptr(da), ffi.ptr(rew),
                                    ffi.ptr(done), ffi.ptr(ds), None))
    torch.cuda.synchronize()
    return dp.cpu().numpy(), dv.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), \
        (ds.cpu().numpy() if ds is not None else None)


def test_mc_step_parity_1mi_rows(ssc):
    from tests.test_device_math_host import _mc_inputs
    pos, vel, act = _mc_inputs()
    p2, v2, rew, done, _ = _mc_step_gpu(ssc, pos, vel, act)
    rp, rv, rr, rd = O.mc_step(pos, vel, act)
    assert np.max(np.abs(p2 - rp)) <= TOL_POS
    assert np.max(np.abs(v2 - rv)) <= TOL_VEL
    clear = np.abs(rp - 0.45) > TOL_POS
    assert np.array_equal(done.astype(bool)[clear], rd[clear])
    r_own = np.where(done.astype(bool), 100.0, 0.0) - act.astype(np.float64) ** 2 * 0.1
    assert np.max(np.abs(rew - r_own) / np.maximum(1.0, np.abs(r_own))) <= 1e-6
    assert v2[0] == 0.0 and p2[0] == np.float32(-1.2) and done[2] == 1 and p2[4] == np.float32(0.6)


def test_mc_step_reference_goldens(ssc, golden_dir):
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    S, A = g["states_val"], g["controls_val"]
    pos = S[:, :-1, 0].reshape(-1).astype(np.float32)
    vel = S[:, :-1, 1].reshape(-1).astype(np.float32)
    act = A[:, :-1, 0].reshape(-1).astype(np.float32)
    p2, v2, rew, done, _ = _mc_step_gpu(ssc, pos, vel, act)
    assert np.max(np.abs(p2 - S[:, 1:, 0].reshape(-1))) <= TOL_POS
    assert np.max(np.abs(v2 - S[:, 1:, 1].reshape(-1))) <= TOL_VEL + 4e-9   # + fp32 rounding of the inputs
    assert not done.any()


def test_mc_step_summary_paths_and_time_limit(ssc, golden_dir):
    """Replays a reference best_path (goal reached after 151 steps) and a 999-step timeout."""
    g = np.load(f"{golden_dir}/mc_summary_paths.npz")
    path = g["f0_best_path"]
    a = ((path[1:, 1] - path[:-1, 1] + 0.0025 * np.cos(3 * path[:-1, 0])) / 0.0015)
    pos, vel = path[:-1, 0].astype(np.float32), path[:-1, 1].astype(np.float32)
    steps = np.arange(len(a), dtype=np.int32)
    p2, v2, rew, done, st = _mc_step_gpu(ssc, pos, vel, a.astype(np.float32), steps=steps)
    assert np.max(np.abs(p2 - path[1:, 0])) <= TOL_POS and np.max(np.abs(v2 - path[1:, 1])) <= TOL_VEL + 4e-9
    assert done[-1] == 1 and not done[:-1].any()
    assert abs(rew.astype(np.float64).sum() - float(g["f0_best_reward"])) < 1e-3
    assert np.array_equal(st, steps + 1)
    # TimeLimit: elapsed 997 -> 998 not done, 998 -> 999 done (stock env), no limit when steps=NULL
    z = np.zeros(2, np.float32)
    _, _, _, d, st = _mc_step_gpu(ssc, z - 0.5, z, z, steps=np.array([997, 998], np.int32))
    assert d.tolist() == [0, 1] and st.tolist() == [998, 999]
    _, _, _, d, _ = _mc_step_gpu(ssc, z - 0.5, z, z)
    assert d.tolist() == [0, 0]


def test_edge_sizes(ssc):
    ffi = ssc._ffi
    p = ffi.default_params(ffi.SSC_ENV_MOUNTAINCAR, 1.0, 999)
    assert ffi.lib().ssc_mc_step(ctypes.byref(p), 0, None, None, None, None, None, None, None) == 0
    for n in (1, 63, 65, 257):
        pos = np.full(n, -0.5, np.float32)
        p2, v2, _, _, _ = _mc_step_gpu(ssc, pos, np.zeros(n, np.float32), np.ones(n, np.float32))
        rp, rv, _, _ = O.mc_step(pos, np.zeros(n), np.ones(n))
        assert np.max(np.abs(p2 - rp)) <= TOL_POS and np.max(np.abs(v2 - rv)) <= TOL_VEL


def test_vecenv_reset_bit_exact_and_masked(ssc):
    n = 1000
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=77, env_id0=5)
    obs = env.reset()
    assert obs.shape == (n, 2)
    ids = np.uint64(5) + np.arange(n, dtype=np.uint64)
    rp, rv = O.mc_reset_state(77, ids, O.RESET_T0)
    assert np.array_equal(obs[:, 0].cpu().numpy(), rp) and np.array_equal(obs[:, 1].cpu().numpy(), rv)
    o2, r, d, info = env.step(torch.zeros(n, 1))
    assert o2.shape == (n, 2) and r.shape == (n,) and d.dtype == torch.bool and info == {}
    mask = torch.zeros(n, dtype=torch.uint8)
    mask[::3] = 1
    o3 = env.reset(mask=mask).cpu().numpy()
    rp2, _ = O.mc_reset_state(77, ids, 0)          # reset after global step 0
    m = mask.numpy().astype(bool)
    assert np.array_equal(o3[m, 0], rp2[m]) and (o3[m, 1] == 0).all()
    assert np.array_equal(o3[~m], o2.cpu().numpy()[~m])
    assert (env.steps.cpu().numpy()[m] == 0).all() and (env.steps.cpu().numpy()[~m] == 1).all()


def _run_rollout(ssc, n, K, seed, env_id0=0, steps0=0, t0=0, max_steps=999, ring_cap=0):
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed, env_id0=env_id0, max_episode_steps=max_steps)
    env.reset()
    env.steps.fill_(steps0)
    env.t = t0
    pos0, vel0 = env.s0.cpu().numpy(), env.s1.cpu().numpy()
    ring = ssc.EpisodeRing(ring_cap, "cuda") if ring_cap else None
    chunk = env.rollout(K, ssc.RandomPolicy(), ring=ring)
    torch.cuda.synchronize()
    return env, chunk, ring, pos0, vel0


def test_rollout_random_teacher_forced_ragged(ssc):
    """Ragged n, unaligned step0, a time-limit reset inside the window (BASELINE config 2 at test size)."""
    n, K, seed, id0, t0 = 4099, 37, 1234, 10**6, 6
    env, chunk, ring, pos0, vel0 = _run_rollout(ssc, n, K, seed, id0, steps0=980, t0=t0, ring_cap=8192)
    res = O.mc_replay_random_rollout(mc_log(chunk), seed, id0, t0, O.mc_power(1.0), 999, pos0, vel0,
                                     np.full(n, 980, np.int64))
    assert_replay_clean(res)
    log = mc_log(chunk)
    # final state == last logged s2 (or the reset draw), elapsed counter as the oracle derives it
    assert np.array_equal(env.steps.cpu().numpy(), res["final_elapsed"])
    last_done = log["done"][-1].astype(bool)
    assert np.array_equal(env.s0.cpu().numpy()[~last_done], log["s2_pos"][-1][~last_done])
    # every env hit the 999-step limit exactly once: K=37 > 999-980
    assert (log["done"].sum(axis=0) == 1).all() and (log["done"][18] == 1).all()
    stats = env.stats.cpu().numpy()
    assert stats[2] == n * K and stats[3] == n
    assert abs(stats[0] - log["rew"].astype(np.float64).sum()) < 1e-2
    (eid, elen, eret), dropped = ring.drain()
    assert dropped == 0 and len(eid) == n
    assert sorted(eid.tolist()) == list(range(id0, id0 + n)) and (elen == 999).all()
    # episode return accumulates from the state's ep_ret (0 after reset) over the 19 steps in this chunk
    order = np.argsort(eid)
    assert np.allclose(eret[order], log["rew"][:19].astype(np.float64).sum(axis=0), atol=1e-4)
    assert np.allclose(env.ep_ret.cpu().numpy(), log["rew"][19:].astype(np.float64).sum(axis=0), atol=1e-4)


def test_rollout_goal_termination_and_reward(ssc):
    """Envs started next to the goal with high speed terminate by reaching it: +100 reward,
    done flag, reset draw, goal counter."""
    n, K, seed = 512, 8, 42
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed)
    env.reset()
    env.s0.fill_(0.43)
    env.s1.fill_(0.06)
    pos0, vel0 = env.s0.cpu().numpy(), env.s1.cpu().numpy()
    chunk = env.rollout(K, ssc.RandomPolicy())
    torch.cuda.synchronize()
    log = mc_log(chunk)
    res = O.mc_replay_random_rollout(log, seed, 0, 0, O.mc_power(1.0), 999, pos0, vel0, np.zeros(n, np.int64))
    assert_replay_clean(res)
    assert (log["done"][0] == 1).all() and (log["rew"][0] > 99.8).all()
    assert env.stats.cpu().numpy()[1] == n


def test_rollout_chunks_compose_and_shard_invariance(ssc):
    """(a) two chunks of K1+K2 == one chunk of K1+K2; (b) splitting the env id range over two
    'ranks' reproduces the same per-env streams bit for bit."""
    n, seed = 1024, 7
    _, whole, _, _, _ = _run_rollout(ssc, n, 40, seed)
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed)
    a = env.rollout(16, ssc.RandomPolicy())
    b = env.rollout(24, ssc.RandomPolicy())
    torch.cuda.synchronize()
    for col in ("obs", "act", "rew", "done", "obs2"):
        w = getattr(whole, col)
        ab = torch.cat([getattr(a, col), getattr(b, col)], dim=-2)
        assert torch.equal(w, ab), col
    halves = [_run_rollout(ssc, n // 2, 40, seed, env_id0=r * (n // 2))[1] for r in range(2)]
    for col in ("obs", "act", "rew", "done", "obs2"):
        w = getattr(whole, col)
        sh = torch.cat([getattr(h, col) for h in halves], dim=-1)
        assert torch.equal(w, sh), col


def test_vecenv_step_api_matches_fused_rollout(ssc):
    """Driving VecEnv.step()/reset(mask) with the policy's actions reproduces the fused kernel's log."""
    n, K, seed = 300, 12, 99
    env, chunk, _, _, _ = _run_rollout(ssc, n, K, seed, steps0=990)
    log = mc_log(chunk)
    env2 = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed)
    obs = env2.reset()
    env2.steps.fill_(990)
    for k in range(K):
        assert np.array_equal(obs.cpu().numpy()[:, 0], log["s_pos"][k])
        obs2, rew, done, _ = env2.step(_dev(log["act"][k]))
        assert np.array_equal(obs2.cpu().numpy()[:, 0], log["s2_pos"][k])
        assert np.array_equal(obs2.cpu().numpy()[:, 1], log["s2_vel"][k])
        assert np.array_equal(rew.cpu().numpy(), log["rew"][k])
        assert np.array_equal(done.cpu().numpy(), log["done"][k].astype(bool))
        obs = env2.reset(mask=done) if done.any() else obs2


def test_rollout_full_size_properties(ssc):
    """BASELINE config 2 at full size (65 536 envs x 1024 steps): size-independent properties
    checked on the device."""
    n, K = 65536, 1024
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=1234)
    chunk = env.rollout(K, ssc.RandomPolicy())
    torch.cuda.synchronize()
    done = chunk.done.bool()
    # continuity: s[k+1] == s2[k] wherever the episode went on
    cont = (chunk.obs[:, 1:] == chunk.obs2[:, :-1]).all(dim=0) | done[:-1]
    assert bool(cont.all())
    assert float(chunk.obs2[0].min()) >= -1.2000001 and float(chunk.obs2[0].max()) <= 0.6000001
    assert float(chunk.obs2[1].abs().max()) <= 0.0700001
    assert float(chunk.act.min()) >= -1.0 and float(chunk.act.max()) < 1.0
    # after a reset the state is a fresh start
    nxt = chunk.obs[:, 1:][:, done[:-1]]
    if nxt.numel():
        assert float(nxt[0].min()) >= -0.6000001 and float(nxt[0].max()) <= -0.3999999 and float(nxt[1].abs().max()) == 0
    stats = env.stats.cpu().numpy()
    assert stats[2] == n * K
    assert stats[3] == int(done.sum().item())
    assert abs(stats[0] - float(chunk.rew.double().sum())) < 1.0
    # 999-step limit: every env finished exactly one timed-out episode unless it reached the goal
    assert int(done[998].sum()) >= n - int(stats[1])
    # reward identity r = 100*goal - 0.1 a^2 with the kernel's own goal flag
    goal = chunk.obs2[0] >= 0.45
    r = goal.float() * 100.0 - 0.1 * chunk.act * chunk.act
    assert float((r - chunk.rew).abs().max()) <= 1e-4


def test_single_env_view_plumbing_config1(ssc):
    """BASELINE config 1: one env, exact gym scalar protocol, 3 episodes through an rlTrain-style
    loop; (len, return) equal the oracle's scalar env driven by the same actions."""
    env = ssc.make("MountainCarContinuous-v0", seed=1234)
    assert env.spec.id == "MountainCarContinuous-v0"
    assert env.action_space.shape == (1,) and env.observation_space.shape == (2,)
    rng = np.random.RandomState(0)
    for ep in range(3):
        obs = env.reset()
        assert isinstance(obs, np.ndarray) and obs.shape == (2,) and obs.dtype == np.float64
        assert -0.6 <= obs[0] <= -0.4 and obs[1] == 0
        ep_len, ret, ret_ref = 0, 0.0, 0.0
        for step in range(1000):
            a = rng.uniform(-1, 1, (1,))
            obs2, r, d, info = env.step(a)
            assert isinstance(r, float) and isinstance(d, bool) and info == {} and obs2.shape == (2,)
            # teacher-forced: the oracle steps from the state the env reported
            p, v, r_ref, d_ref = O.mc_step_scalar(float(obs[0]), float(obs[1]), float(np.float32(a[0])))
            assert abs(obs2[0] - p) <= TOL_POS and abs(obs2[1] - v) <= TOL_VEL
            ep_len += 1
            ret += r
            ret_ref += r_ref
            if d:
                break
            obs = obs2
        assert ep_len == 999 or obs2[0] >= 0.45
        assert abs(ret - ret_ref) < 1e-2


def test_pack_transitions_kernel_matches_layout(ssc):
    """ssc_pack_transitions == the column-by-column pack of sharding.TransitionGather (the layout the
    world-2 gloo test exercises on the CPU)."""
    from smartstartcontinuous_amd.sharding import TransitionGather
    # n % 4 == 0 takes the 16-byte path, 257 / 1002 the scalar one; dense and packed chunk layouts
    for env_name, n, K, g, packed in [("MountainCarContinuous-v0", 1000, 24, 5, True), ("Pendulum-v0", 257, 8, 8, True),
                                      ("MountainCarContinuous-v0", 1002, 9, 3, True), ("Pendulum-v0", 4096, 6, 2, False),
                                      ("MountainCarContinuous-v0", 65536, 4, 4, True)]:
        env = ssc.VecEnv(env_name, n, seed=3)
        chunk = env.rollout(K, ssc.RandomPolicy(), out=ssc.TransitionChunk(env.obs_dim, K, n, "cuda", packed=packed))
        tg = TransitionGather(env.obs_dim, g, n, 1, 0, "cuda")
        tg.pack(chunk, 0, env.stats)
        torch.cuda.synchronize()
        obs, act, rew, obs2, done = tg._views(tg.send[0])
        assert torch.equal(obs, chunk.obs[:, K - g:]) and torch.equal(obs2, chunk.obs2[:, K - g:])
        assert torch.equal(act, chunk.act[K - g:]) and torch.equal(rew, chunk.rew[K - g:])
        assert torch.equal(done, chunk.done[K - g:])
        assert torch.equal(tg._stats_view(tg.send[0]), env.stats)
