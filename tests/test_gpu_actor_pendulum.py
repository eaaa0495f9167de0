"""GPU parity tests: DDPG actor forward (fp32 VALU + bf16 MFMA), the fused actor rollout
(BASELINE config 3) and the Pendulum kernels, all through the C ABI and against the oracle."""
import ctypes

import numpy as np
import pytest

from oracle import ssc_oracle as O
from tests.gpu_util import actor_weights

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL_ACT_F32 = 1e-5     # fp32 kernel vs fp64 oracle, abs on actions
TOL_ACT_BF16 = 2e-2    # bf16 MFMA kernel vs fp32/fp64
TOL_ACT_BF16_EMU = 1.5e-3  # vs the oracle's own bf16 emulation: equal except where a hidden activation sits on a
                         # bf16 rounding boundary and fp32/fp64 evaluation rounds it to different neighbours


@pytest.fixture(scope="module")
def ssc():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    import smartstartcontinuous_amd as pkg
    pkg._ffi.lib()
    return pkg


def _actor_forward_gpu(ssc, w, obs, precision, last_layer_tanh=True, act_dim=1):
    ffi = ssc._ffi
    d = {k: torch.as_tensor(v, device="cuda").contiguous() for k, v in w.items()}
    a = ffi.ActorDesc()
    a.obs_dim, a.h1 = w["W1"].shape
    a.h2, a.act_dim = w["W2"].shape[1], act_dim
    a.W1, a.b1, a.W2, a.b2, a.W3, a.b3 = (d[k].data_ptr() for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
    a.last_layer_tanh, a.precision = int(last_layer_tanh), precision
    o = torch.as_tensor(obs, dtype=torch.float32, device="cuda").contiguous()
    out = torch.empty((o.shape[0], act_dim), dtype=torch.float32, device="cuda")
    ffi.check(ffi.lib().ssc_actor_forward(ctypes.byref(a), o.shape[0], ffi.ptr(o), ffi.ptr(out), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("obs_dim", [2, 3])
@pytest.mark.parametrize("llt", [True, False])
def test_actor_forward_f32_and_mfma(ssc, obs_dim, llt):
    ffi = ssc._ffi
    rng = np.random.default_rng(10 + obs_dim)
    for (h1, h2) in [(64, 32), (128, 64), (48, 24), (200, 100), (160, 96)]:   # 200-100: W2 fragments staged in LDS
        w = actor_weights(obs_dim, h1, h2, seed=h1 + obs_dim, w3_scale=0.3)
        for m in (1, 63, 1000, 4096 + 17):
            obs = rng.uniform(-1.2, 1.2, size=(m, obs_dim)).astype(np.float32)
            ref = O.actor_forward(obs, **w, last_layer_tanh=llt)
            got = _actor_forward_gpu(ssc, w, obs, ffi.SSC_PREC_F32, llt)
            assert np.max(np.abs(got - ref)) <= TOL_ACT_F32, (h1, h2, m)
            got_b = _actor_forward_gpu(ssc, w, obs, ffi.SSC_PREC_BF16_MFMA, llt)
            assert np.max(np.abs(got_b - ref)) <= TOL_ACT_BF16, (h1, h2, m)
            emu = O.actor_forward_bf16emu(obs, **w, last_layer_tanh=llt)
            assert np.max(np.abs(got_b - emu)) <= TOL_ACT_BF16_EMU, (h1, h2, m)


def test_actor_forward_generic_sizes(ssc):
    """200-100 (hidden_layer_size_experiment) and multi-action heads through the generic fp32 kernel."""
    ffi = ssc._ffi
    rng = np.random.default_rng(5)
    w = actor_weights(2, 200, 100, seed=3, w3_scale=0.2)
    obs = rng.uniform(-1, 1, size=(777, 2)).astype(np.float32)
    got = _actor_forward_gpu(ssc, w, obs, ffi.SSC_PREC_F32)
    assert np.max(np.abs(got - O.actor_forward(obs, **w))) <= TOL_ACT_F32
    # a handful of rows (the scalar get_action path) run one block per row: same sums in the same order, identical bits
    few = _actor_forward_gpu(ssc, w, obs[:20], ffi.SSC_PREC_F32)
    assert np.array_equal(few, got[:20])
    w = actor_weights(4, 64, 32, seed=4, w3_scale=0.2)
    w["W3"] = rng.uniform(-0.2, 0.2, size=(32, 3)).astype(np.float32)
    w["b3"] = rng.uniform(-0.1, 0.1, size=3).astype(np.float32)
    obs = rng.uniform(-1, 1, size=(130, 4)).astype(np.float32)
    got = _actor_forward_gpu(ssc, w, obs, ffi.SSC_PREC_F32, act_dim=3)
    assert got.shape == (130, 3) and np.max(np.abs(got - O.actor_forward(obs, **w))) <= TOL_ACT_F32


def test_mfma_layout_with_asymmetric_weights(ssc):
    """A=I-style check: one-hot weights expose any row/col or k-permutation mix-up exactly."""
    ffi = ssc._ffi
    h1, h2 = 64, 32
    for trial in range(4):
        rng = np.random.default_rng(trial)
        w = dict(W1=np.zeros((2, h1), np.float32), b1=np.zeros(h1, np.float32), W2=np.zeros((h1, h2), np.float32),
                 b2=np.zeros(h2, np.float32), W3=np.zeros((h2, 1), np.float32), b3=np.zeros(1, np.float32))
        u, j = int(rng.integers(h1)), int(rng.integers(h2))
        w["W1"][trial % 2, u] = 1.0          # hidden unit u copies obs component
        w["W2"][u, j] = 0.5                  # exactly representable in bf16
        w["W3"][j, 0] = 1.0
        obs = rng.uniform(0.1, 1.0, size=(64 * 3, 2)).astype(np.float32)
        got = _actor_forward_gpu(ssc, w, obs, ffi.SSC_PREC_BF16_MFMA, last_layer_tanh=False)
        ref = np.tanh(0.5 * O.round_bf16(obs[:, trial % 2]).astype(np.float64))[:, None]
        assert np.max(np.abs(got - ref)) < 1e-6, (trial, u, j)


def _log(chunk):
    return dict(obs=chunk.obs.cpu().numpy(), act=chunk.act.cpu().numpy(), rew=chunk.rew.cpu().numpy(),
                done=chunk.done.cpu().numpy(), obs2=chunk.obs2.cpu().numpy())


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("bf16_mfma", TOL_ACT_BF16)])
def test_rollout_actor_mountaincar_teacher_forced(ssc, precision, tol):
    """BASELINE config 3 at test size: actor 64-32 + OU noise fused with the step."""
    n, K, seed, id0 = 1000, 50, 1234, 77
    w = actor_weights(2, 64, 32, seed=1234, w3_scale=0.5)
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed, env_id0=id0)
    obs0 = env.reset().cpu().numpy()
    env.steps.fill_(970)           # time-limit reset (and OU reset) inside the window
    env.t = 3                      # odd, unaligned start
    pol = ssc.ActorPolicy({k: torch.as_tensor(v) for k, v in w.items()}, precision=precision)
    chunk = env.rollout(K, pol)
    torch.cuda.synchronize()
    oracle_pol = O.OracleDDPGPolicy(w, seed, id0, n, bf16=False)
    res = O.replay_rollout("mc", _log(chunk), seed, id0, 3, 999, obs0, np.full(n, 970), oracle_pol)
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0, res
    assert res["max_dact"] <= tol, res
    assert res["max_dobs2"][0] <= 2.4e-7 and res["max_dobs2"][1] <= 1e-8, res
    assert res["max_drew_rel"] <= 1e-6 and res["reset_max_err"] == 0, res
    # OU state written back == the oracle's (fp64) state
    assert np.max(np.abs(env.ou_x.cpu().numpy() - oracle_pol.x)) < 2e-5
    if precision == "bf16_mfma":
        emu = O.OracleDDPGPolicy(w, seed, id0, n, bf16=True)
        res = O.replay_rollout("mc", _log(chunk), seed, id0, 3, 999, obs0, np.full(n, 970), emu)
        assert res["max_dact"] <= TOL_ACT_BF16_EMU, res


@pytest.mark.parametrize("env_id,h1,h2", [("MountainCarContinuous-v0", 200, 100), ("MountainCarContinuous-v0", 128, 64),
                                          ("Pendulum-v0", 200, 100)])
def test_rollout_actor_wide_shapes_teacher_forced(ssc, env_id, h1, h2):
    """The reference's actor grid beyond 64-32 (data/ddpg_baselines_summaries/hidden_layer_size_experiment/: 128-64 and
    200-100) in the FUSED rollout: 128-64 keeps its W2 fragments in registers, 200-100 stages them in LDS (ActorMfmaLds)."""
    n, K, seed, id0 = 700, 24, 99, 11
    pend = env_id.startswith("Pendulum")
    obs_dim = 3 if pend else 2
    w = actor_weights(obs_dim, h1, h2, seed=h1, w3_scale=0.4)
    env = ssc.VecEnv(env_id, n, seed=seed, env_id0=id0)
    obs0 = env.reset().cpu().numpy()
    start = 190 if pend else 990
    env.steps.fill_(start)         # time-limit reset (and OU reset) inside the window
    env.t = 5
    pol = ssc.ActorPolicy({k: torch.as_tensor(v) for k, v in w.items()}, precision="bf16_mfma", obs_clip=5.0)
    chunk = env.rollout(K, pol)
    torch.cuda.synchronize()
    low, high = (-2.0, 2.0) if pend else (-1.0, 1.0)
    oracle_pol = O.OracleDDPGPolicy(w, seed, id0, n, bf16=False, low=low, high=high, obs_clip=5.0)
    res = O.replay_rollout("pend" if pend else "mc", _log(chunk), seed, id0, 5, 200 if pend else 999, obs0,
                           np.full(n, start), oracle_pol)
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0, res
    assert res["max_dact"] <= TOL_ACT_BF16 * (high - low) / 2, res
    emu = O.OracleDDPGPolicy(w, seed, id0, n, bf16=True, low=low, high=high, obs_clip=5.0)
    res = O.replay_rollout("pend" if pend else "mc", _log(chunk), seed, id0, 5, 200 if pend else 999, obs0,
                           np.full(n, start), emu)
    assert res["max_dact"] <= TOL_ACT_BF16_EMU * (high - low) / 2 * 2, res


@pytest.mark.parametrize("llt", [True, False])
def test_rollout_actor_chunks_compose(ssc, llt):
    """The fused actor rollout must not depend on how the steps were cut into launches: chunks of 7 + 1 + 18 + 14
    steps from an unaligned start reproduce ONE chunk of 40 steps bit for bit (the step body exists in head / main-loop
    / tail copies inside the kernel, and one Philox call serves 4 steps)."""
    n, seed = 2000, 11
    w = {k: torch.as_tensor(v) for k, v in actor_weights(2, 64, 32, seed=3, w3_scale=0.5).items()}

    def fresh():
        env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed, env_id0=5)
        env.reset()
        env.steps.fill_(985)
        env.t = 2
        return env
    pol = ssc.ActorPolicy(w, precision="bf16_mfma", last_layer_tanh=llt)
    env = fresh()
    whole = env.rollout(40, pol)
    env2 = fresh()
    parts = [env2.rollout(k, pol) for k in (7, 1, 18, 14)]
    torch.cuda.synchronize()
    for col in ("obs", "act", "rew", "done", "obs2"):
        assert torch.equal(getattr(whole, col), torch.cat([getattr(p, col) for p in parts], dim=-2)), col
    assert torch.equal(env.ou_x, env2.ou_x) and torch.equal(env.s0, env2.s0) and torch.equal(env.stats[1:], env2.stats[1:])
    assert abs(float(env.stats[0] - env2.stats[0])) < 1e-2          # per-launch fp32 partial sums of the rewards


def test_rollout_actor_without_noise_equals_actor_forward(ssc):
    n, K = 320, 9
    w = actor_weights(2, 64, 32, seed=9, w3_scale=0.5)
    env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=5)
    env.reset()
    pol = ssc.ActorPolicy({k: torch.as_tensor(v) for k, v in w.items()}, precision="f32", ou_epsilon=0.0)
    chunk = env.rollout(K, pol)
    torch.cuda.synchronize()
    obs = chunk.obs.cpu().numpy().transpose(1, 2, 0).reshape(-1, 2)
    ref = np.clip(O.actor_forward(obs, **w)[:, 0], -1, 1)
    assert np.max(np.abs(chunk.act.cpu().numpy().reshape(-1) - ref)) <= TOL_ACT_F32


# ------------------------------------------------------------------------- Pendulum --
def test_pendulum_step_kernel(ssc):
    ffi = ssc._ffi
    rng = np.random.default_rng(8)
    n = 1 << 18
    th = rng.uniform(-30, 30, n).astype(np.float32)
    thd = rng.uniform(-8, 8, n).astype(np.float32)
    act = rng.uniform(-3, 3, n).astype(np.float32)
    for v1 in (0, 1):
        p = ffi.default_params(ffi.SSC_ENV_PENDULUM, 1.0, 200)
        p.pend_v1_order = v1
        dt, dd, da = (torch.as_tensor(x.copy(), device="cuda") for x in (th, thd, act))
        obs = torch.empty((3, n), dtype=torch.float32, device="cuda")
        rew = torch.empty(n, dtype=torch.float32, device="cuda")
        done = torch.empty(n, dtype=torch.uint8, device="cuda")
        steps = torch.full((n,), 198, dtype=torch.int32, device="cuda")
        steps[: n // 2] = 199
        ffi.check(ffi.lib().ssc_pend_step(ctypes.byref(p), n, ffi.ptr(dt), ffi.ptr(dd), ffi.ptr(da), ffi.ptr(obs),
                                          ffi.ptr(rew), ffi.ptr(done), ffi.ptr(steps), None))
        torch.cuda.synchronize()
        rt, rd, rr, _ = O.pend_step(th, thd, act, v1_order=bool(v1))
        assert np.max(np.abs(dt.cpu().numpy() - rt)) <= 4e-6
        assert np.max(np.abs(dd.cpu().numpy() - rd)) <= 2e-6
        assert np.max(np.abs(rew.cpu().numpy() - rr) / np.maximum(1, np.abs(rr))) <= 2e-5
        o = obs.cpu().numpy()
        assert np.max(np.abs(o[0] - np.cos(rt))) <= 5e-6 and np.max(np.abs(o[1] - np.sin(rt))) <= 5e-6
        assert np.array_equal(o[2], dd.cpu().numpy())
        d = done.cpu().numpy()
        assert d[: n // 2].all() and not d[n // 2:].any()       # TimeLimit(200) only


@pytest.mark.parametrize("env_id,v1", [("Pendulum-v0", False), ("Pendulum-v1", True)])
def test_pendulum_env_id_selects_update_order(ssc, env_id, v1):
    """The env built BY ID runs the update order its id names (gym 0.10.5 'v0': theta integrates the unclipped new
    velocity; modern 'v1', the id BASELINE configs[3] uses: clip first) -- through VecEnv.step and through the fused
    rollout, with a third of the envs at |theta-dot| close to 8 so that the two orders differ."""
    n, K, seed, id0 = 1536, 24, 99, 11
    env = ssc.VecEnv(env_id, n, seed=seed, env_id0=id0)
    assert env.params.pend_v1_order == int(v1) and env.spec.id == env_id
    env.reset()
    env.s1[: n // 3] = torch.where(env.s1[: n // 3] >= 0, 7.9, -7.9)
    # single-step API
    th, thd = env.s0.cpu().numpy().copy(), env.s1.cpu().numpy().copy()
    act = np.where(thd >= 0, 2.0, -2.0).astype(np.float32)
    env.step(torch.as_tensor(act))
    torch.cuda.synchronize()
    rt, rd, _, _ = O.pend_step(th, thd, act, v1_order=v1)
    wt, _, _, _ = O.pend_step(th, thd, act, v1_order=not v1)
    assert np.max(np.abs(env.s0.cpu().numpy() - rt)) <= 4e-6 and np.max(np.abs(env.s1.cpu().numpy() - rd)) <= 2e-6
    assert np.max(np.abs(wt - rt)) > 1e-3                        # the case separates the two orders
    # fused rollout, teacher forced
    env.s1[: n // 3] = torch.where(env.s1[: n // 3] >= 0, 7.9, -7.9)
    obs0 = env.observe().cpu().numpy().copy()
    steps0 = env.steps.cpu().numpy().astype(np.int64)
    t0 = env.t
    chunk = env.rollout(K, ssc.RandomPolicy())
    torch.cuda.synchronize()
    pol = O.OracleRandomPolicy(seed, id0, n, -2.0, 2.0)
    res = O.replay_rollout("pend", _log(chunk), seed, id0, t0, 200, obs0, steps0, pol, v1_order=v1)
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0, res
    assert res["max_dact"] == 0 and (res["max_dobs2"] <= [3e-6, 3e-6, 3e-6]).all(), res
    wrong = O.replay_rollout("pend", _log(chunk), seed, id0, t0, 200, obs0, steps0,
                             O.OracleRandomPolicy(seed, id0, n, -2.0, 2.0), v1_order=not v1)
    assert wrong["max_dobs2"][:2].max() > 1e-3, wrong


@pytest.mark.parametrize("policy", ["random", "actor_f32", "actor_mfma", "actor_f32_clip", "actor_mfma_clip"])
def test_rollout_pendulum_teacher_forced(ssc, policy):
    n, K, seed, id0 = 777, 40, 4321, 5
    env = ssc.VecEnv("Pendulum-v0", n, seed=seed, env_id0=id0)
    assert env.obs_dim == 3 and env.spec.max_episode_steps == 200 and env.action_space.high[0] == 2.0
    obs0 = env.reset().cpu().numpy()
    rt, rd = O.pend_reset_state(seed, np.uint64(id0) + np.arange(n, dtype=np.uint64), O.RESET_T0)
    assert np.max(np.abs(obs0[:, 0] - np.cos(rt.astype(np.float64)))) < 2e-6 and np.array_equal(obs0[:, 2], rd)
    env.steps.fill_(180)
    env.t = 2
    if policy == "random":
        pol, oracle_pol, tol = ssc.RandomPolicy(), O.OracleRandomPolicy(seed, id0, n, -2.0, 2.0), 0.0
    else:
        w = actor_weights(3, 64, 32, seed=21, w3_scale=0.5)
        prec = "f32" if policy.startswith("actor_f32") else "bf16_mfma"
        # "_clip": what DDPG_Baselines_agent.as_policy() builds -- the actor sees clip(obs, -5, 5) (ddpg_editted.py:106-109);
        # a quarter of the envs start with |theta-dot| > 5 so that the clip is active
        clip = 5.0 if policy.endswith("_clip") else 0.0
        if clip:
            env.s1[: n // 4] = torch.where(env.s1[: n // 4] >= 0, 7.0, -7.0)
            obs0 = env.observe().cpu().numpy().copy()
        pol = ssc.ActorPolicy({k: torch.as_tensor(v) for k, v in w.items()}, precision=prec, obs_clip=clip)
        oracle_pol = O.OracleDDPGPolicy(w, seed, id0, n, low=-2.0, high=2.0, obs_clip=clip or None)
        tol = 4e-5 if prec == "f32" else 2 * TOL_ACT_BF16
    chunk = env.rollout(K, pol)
    torch.cuda.synchronize()
    res = O.replay_rollout("pend", _log(chunk), seed, id0, 2, 200, obs0, np.full(n, 180), oracle_pol)
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0, res
    assert res["max_dact"] <= tol, res
    assert (res["max_dobs2"] <= [3e-6, 3e-6, 3e-6]).all(), res
    assert res["max_drew_rel"] <= 2e-5 and res["reset_max_err"] <= 2e-6, res
    log = _log(chunk)
    assert (log["done"][19] == 1).all() and log["done"].sum() == n           # 180 + 20 = 200
    assert abs(env.stats.cpu().numpy()[0] - log["rew"].astype(np.float64).sum()) < 0.5


def test_rollout_actor_full_size_properties(ssc):
    """BASELINE config 3 at full size (65 536 envs x 1024 steps -- the chunk the bench leg launches --, actor 64-32 on the bf16 MFMA + OU noise):
    size-independent properties on the device, the logged action equal to the standalone actor kernel plus a
    noise term bounded by the OU process, bit-for-bit repeatability, and a 128-env slice replayed by the oracle."""
    n, K, seed = 65536, 1024, 1234
    w = actor_weights(2, 64, 32, seed=1234, w3_scale=0.5)
    wt = {k: torch.as_tensor(v) for k, v in w.items()}

    def run():
        env = ssc.VecEnv("MountainCarContinuous-v0", n, seed=seed)
        obs0 = env.reset().cpu().numpy()
        chunk = env.rollout(K, ssc.ActorPolicy(wt, precision="bf16_mfma"))
        torch.cuda.synchronize()
        return env, obs0, chunk
    env, obs0, chunk = run()
    done = chunk.done.bool()
    cont = (chunk.obs[:, 1:] == chunk.obs2[:, :-1]).all(dim=0) | done[:-1]
    assert bool(cont.all())
    assert float(chunk.act.abs().max()) <= 1.0                                   # clip of DDPG_editted.pi (:271)
    assert float(chunk.obs2[0].min()) >= -1.2000001 and float(chunk.obs2[0].max()) <= 0.6000001
    assert float(chunk.obs2[1].abs().max()) <= 0.0700001
    goal = chunk.obs2[0] >= 0.45
    r = goal.float() * 100.0 - 0.1 * chunk.act * chunk.act
    assert float((r - chunk.rew).abs().max()) <= 1e-4
    # done = goal or TimeLimit(999): steps since the env's previous done (or the chunk's start), on the device
    idx = torch.arange(K, device=done.device, dtype=torch.int32)[:, None].expand(K, n)
    last = torch.cummax(torch.where(done, idx, torch.full_like(idx, -1)), dim=0).values
    prev = torch.cat([torch.full_like(last[:1], -1), last[:-1]])                   # index of the previous done, -1: none yet
    elapsed = idx - prev
    assert bool((done == (goal | (elapsed == 999))).all()) and int((done & ~goal).sum()) > 0
    stats = env.stats.cpu().numpy()
    assert stats[2] == n * K and stats[3] == int(done.sum().item())
    # same seed, same ids -> the same bits
    _, _, chunk2 = run()
    assert torch.equal(chunk.act, chunk2.act) and torch.equal(chunk.obs2, chunk2.obs2) and torch.equal(chunk.rew, chunk2.rew)
    # a slice of envs, every step re-derived by the oracle from the logged state (teacher forced)
    sl = slice(4000, 4128)
    log = {k: v[..., sl] for k, v in _log(chunk).items()}
    pol = O.OracleDDPGPolicy(w, seed, 4000, 128, bf16=False)
    res = O.replay_rollout("mc", log, seed, 4000, 0, 999, obs0[sl], np.zeros(128, np.int64), pol)
    assert res["start_max_err"] == 0 and res["continuity_mismatch"] == 0 and res["done_mismatch"] == 0, res
    assert res["max_dact"] <= TOL_ACT_BF16 and res["max_drew_rel"] <= 1e-6, res
    assert res["max_dobs2"][0] <= 2.4e-7 and res["max_dobs2"][1] <= 1e-8, res
