"""The per-env device functions (csrc/ssc_device.h), compiled for the host by hipcc and run
on the CPU, against the fp64 oracle: RNG bit-exactness and the fp32 error budget of the
step (SURVEY.md section 8d config-2 tolerances).  No GPU needed."""
import ctypes
import os
import shutil
import sys

import numpy as np
import pytest

from oracle import ssc_oracle as O

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                                reason="hipcc not available")

TOL_VEL, TOL_POS = 1e-8, 2.4e-7


@pytest.fixture(scope="module")
def H():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "host_harness"))
    import build as hb
    return ctypes.CDLL(hb.build())


@pytest.fixture(scope="module")
def mc_params():
    from smartstartcontinuous_amd import _ffi
    return _ffi.default_params(_ffi.SSC_ENV_MOUNTAINCAR, 1.0, 999)


def fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_philox_and_keying_bit_exact(H):
    out = np.zeros(4, np.uint32)
    ctr = np.array([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], np.uint32)
    key = np.array([0xa4093822, 0x299f31d0], np.uint32)
    H.h_philox(fp(ctr), fp(key), fp(out))
    assert [hex(x) for x in out] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']
    for seed, env, t, tag in [(1234, 0, 0, 0), (2**40 + 17, 2**33 + 5, 2**34 + 9, 3), (1, 65535, O.RESET_T0, 1)]:
        H.h_rng_words(ctypes.c_uint64(seed), ctypes.c_uint64(env), ctypes.c_uint64(t), ctypes.c_uint32(tag), fp(out))
        ref = O.rng_words(seed, np.uint64(env), np.uint64(t), tag)
        assert [int(x) for x in out] == [int(r) for r in ref]


def test_uniform_bit_exact(H):
    rng = np.random.default_rng(0)
    x = rng.integers(0, 2**32, size=200000, dtype=np.uint64).astype(np.uint32)
    x[:4] = [0, 0xFFFFFFFF, 0xFF, 0x100]
    out = np.empty(x.size, np.float32)
    for low, high in [(-1.0, 1.0), (-0.6, -0.4), (-2.0, 2.0), (-np.pi, np.pi)]:
        low32 = np.float32(low)
        span = np.float32(np.float32(high) - low32)
        H.h_uniform(ctypes.c_int64(x.size), fp(x), ctypes.c_float(low32), ctypes.c_float(span), fp(out))
        assert np.array_equal(out, O.uniform_f32(x, low, high))
        assert out.min() >= low32 and out.max() < np.float32(high) + 1e-6


def test_cos_bounded_error(H):
    x = np.linspace(-4.71, 4.71, 400001).astype(np.float32)
    out = np.empty_like(x)
    H.h_cos_bounded(ctypes.c_int64(x.size), fp(x), fp(out))
    err = np.abs(out.astype(np.float64) - np.cos(x.astype(np.float64)))
    assert err.max() < 1.5e-7


def _mc_inputs(n=1 << 20):
    """SURVEY.md 8d parity inputs: 1 Mi rows + hand KATs."""
    rng = np.random.default_rng(1234)
    pos = rng.uniform(-1.2, 0.6, n).astype(np.float32)
    vel = rng.uniform(-0.07, 0.07, n).astype(np.float32)
    act = rng.uniform(-1.5, 1.5, n).astype(np.float32)
    kat = np.array([
        # pos, vel, act
        [-1.2, -0.01, -1.0],      # at the left wall moving left -> vel zeroed
        [-1.1999, -0.07, -1.0],   # hits the wall this step
        [0.44, 0.07, 1.0],        # crosses the goal
        [0.449, 0.0009, 1.0],     # lands next to the goal threshold
        [0.59, 0.07, 1.0],        # right clamp
        [-0.5, 0.0699, 1.0],      # velocity clamp +
        [-0.5, -0.0699, -1.0],    # velocity clamp -
        [-0.5, 0.0, 7.5],         # |a| > 1: force clipped, reward uses raw action
        [-0.5, 0.0, -7.5],
        [0.0, 0.0, 0.0],
    ], np.float32)
    k = len(kat)
    pos[:k], vel[:k], act[:k] = kat[:, 0], kat[:, 1], kat[:, 2]
    return pos, vel, act


def test_mc_step_fp32_vs_oracle(H, mc_params):
    pos, vel, act = _mc_inputs()
    n = pos.size
    p2, v2 = pos.copy(), vel.copy()
    rew = np.empty(n, np.float32)
    goal = np.empty(n, np.uint8)
    H.h_mc_step(ctypes.byref(mc_params), ctypes.c_int64(n), fp(p2), fp(v2), fp(act), fp(rew), fp(goal))
    rp, rv, rr, rd = O.mc_step(pos, vel, act)
    assert np.max(np.abs(p2 - rp)) <= TOL_POS
    assert np.max(np.abs(v2 - rv)) <= TOL_VEL
    clear = np.abs(rp - 0.45) > TOL_POS
    assert np.array_equal(goal.astype(bool)[clear], rd[clear])
    # reward is consistent with the function's own goal flag
    r_own = np.where(goal.astype(bool), 100.0, 0.0) - act.astype(np.float64) ** 2 * 0.1
    assert np.max(np.abs(rew - r_own) / np.maximum(1.0, np.abs(r_own))) <= 1e-6
    # hand KATs
    assert v2[0] == 0.0 and p2[0] == np.float32(-1.2)
    assert v2[1] == 0.0 and p2[1] == np.float32(-1.2)
    assert goal[2] == 1 and rew[2] > 99.0
    assert p2[4] == np.float32(0.6)
    assert v2[5] == np.float32(0.07) and v2[6] == np.float32(-0.07)
    assert abs(rew[7] - (-0.1 * 7.5 ** 2)) < 1e-5 and abs(v2[7] - rv[7]) <= TOL_VEL


def test_mc_step_on_reference_goldens(H, mc_params, golden_dir):
    """The reference's own recorded transitions through the fp32 device function."""
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    S, A = g["states_val"], g["controls_val"]
    pos = S[:, :-1, 0].reshape(-1).astype(np.float32)
    vel = S[:, :-1, 1].reshape(-1).astype(np.float32)
    act = A[:, :-1, 0].reshape(-1).astype(np.float32)
    n = pos.size
    p2, v2 = pos.copy(), vel.copy()
    rew = np.empty(n, np.float32)
    goal = np.empty(n, np.uint8)
    H.h_mc_step(ctypes.byref(mc_params), ctypes.c_int64(n), fp(p2), fp(v2), fp(act), fp(rew), fp(goal))
    # inputs were rounded to fp32 (<= 6e-8 on pos, 4e-9 on vel): budget = tolerance + input rounding
    assert np.max(np.abs(p2 - S[:, 1:, 0].reshape(-1))) <= TOL_POS
    assert np.max(np.abs(v2 - S[:, 1:, 1].reshape(-1))) <= TOL_VEL + 4e-9
    assert not goal.any()


def test_mc_reset_bit_exact(H, mc_params):
    n = 4097
    pos = np.empty(n, np.float32)
    vel = np.empty(n, np.float32)
    H.h_mc_reset(ctypes.byref(mc_params), ctypes.c_int64(n), ctypes.c_uint64(1234), ctypes.c_uint64(77),
                 ctypes.c_uint64(O.RESET_T0), fp(pos), fp(vel))
    rp, rv = O.mc_reset_state(1234, np.uint64(77) + np.arange(n, dtype=np.uint64), O.RESET_T0)
    assert np.array_equal(pos, rp) and np.array_equal(vel, rv)
    assert pos.min() >= np.float32(-0.6) and pos.max() <= np.float32(-0.4)


def test_pendulum_step_fp32_vs_oracle(H):
    from smartstartcontinuous_amd import _ffi
    rng = np.random.default_rng(5)
    n = 1 << 18
    th = rng.uniform(-30, 30, n).astype(np.float32)
    thd = rng.uniform(-8, 8, n).astype(np.float32)
    act = rng.uniform(-3, 3, n).astype(np.float32)
    for v1 in (0, 1):
        p = _ffi.default_params(_ffi.SSC_ENV_PENDULUM, 1.0, 200)
        p.pend_v1_order = v1
        t2, d2 = th.copy(), thd.copy()
        rew = np.empty(n, np.float32)
        H.h_pend_step(ctypes.byref(p), ctypes.c_int64(n), fp(t2), fp(d2), fp(act), fp(rew))
        rt, rd, rr, _ = O.pend_step(th, thd, act, v1_order=bool(v1))
        assert np.max(np.abs(t2 - rt)) <= 4e-6          # 1 ulp at |th| ~ 30 is 1.9e-6
        assert np.max(np.abs(d2 - rd)) <= 2e-6
        assert np.max(np.abs(rew - rr) / np.maximum(1.0, np.abs(rr))) <= 2e-5
    th_r = np.empty(100, np.float32)
    thd_r = np.empty(100, np.float32)
    H.h_pend_reset(ctypes.c_int64(100), ctypes.c_uint64(9), ctypes.c_uint64(0), ctypes.c_uint64(3), fp(th_r), fp(thd_r))
    rt, rd = O.pend_reset_state(9, np.arange(100, dtype=np.uint64), 3)
    assert np.array_equal(th_r, rt) and np.array_equal(thd_r, rd)


def test_angle_normalize_and_tanh(H):
    x = np.linspace(-100, 100, 200001).astype(np.float32)
    out = np.empty_like(x)
    H.h_angle_normalize(ctypes.c_int64(x.size), fp(x), fp(out))
    ref = O.angle_normalize(x.astype(np.float64))
    d = np.abs(out - ref)
    d = np.minimum(d, np.abs(d - 2 * np.pi))        # the wrap point itself may land on either side
    assert d.max() < 2e-5
    assert out.min() >= -np.pi - 1e-6 and out.max() <= np.pi + 1e-6
    x = np.linspace(-20, 20, 100001).astype(np.float32)
    H.h_tanh_fast(ctypes.c_int64(x.size), fp(x), fp(out[:x.size]))
    assert np.max(np.abs(out[:x.size] - np.tanh(x.astype(np.float64)))) < 3e-7
