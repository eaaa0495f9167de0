"""The CPU oracle against the reference's own recorded data.

These are the pins that make oracle/ssc_oracle.py trustworthy for MountainCar and for
the geometry helpers; everything runs on CPU."""
import ctypes

import numpy as np
import pytest

from oracle import ssc_oracle as O


@pytest.fixture(scope="module")
def rollouts(golden_dir):
    return np.load(f"{golden_dir}/mc_reference_rollouts.npz")


@pytest.fixture(scope="module")
def summaries(golden_dir):
    return np.load(f"{golden_dir}/mc_summary_paths.npz")


@pytest.fixture(scope="module")
def kats(golden_dir):
    return np.load(f"{golden_dir}/numerical_kats.npz")


def test_mc_step_reproduces_reference_validation_rollouts(rollouts):
    """states_val/controls_val: 20 x 333 recorded fp64 transitions of the reference env
    (models/NND_MB_agent/default/training_data) -- exact to rounding."""
    S, A = rollouts["states_val"], rollouts["controls_val"]
    p2, v2, r, d = O.mc_step(S[:, :-1, 0], S[:, :-1, 1], A[:, :-1, 0])
    assert np.max(np.abs(p2 - S[:, 1:, 0])) < 1e-15
    assert np.max(np.abs(v2 - S[:, 1:, 1])) < 1e-15
    assert not d.any()


def test_mc_step_reproduces_reference_training_deltas(rollouts):
    """dataX/dataY/dataZ: 8300 (s, a, s'-s) rows."""
    X, Y, Z = rollouts["dataX"], rollouts["dataY"], rollouts["dataZ"]
    p2, v2, _, _ = O.mc_step(X[:, 0], X[:, 1], Y[:, 0])
    assert np.max(np.abs((p2 - X[:, 0]) - Z[:, 0])) < 1e-15
    assert np.max(np.abs((v2 - X[:, 1]) - Z[:, 1])) < 1e-15


def test_mc_step_reproduces_forwardsim_fixture(rollouts):
    X, Y = rollouts["forwardsim_x_true"], rollouts["forwardsim_y"]
    p2, v2, _, _ = O.mc_step(X[:-1, 0], X[:-1, 1], Y[:-1, 0])
    assert np.max(np.abs(p2 - X[1:, 0])) < 1e-15
    assert np.max(np.abs(v2 - X[1:, 1])) < 1e-15


def _actions_from_path(path, power):
    """Invert the velocity update for the applied force (valid where no clamp was active)."""
    p, v = path[:-1, 0], path[:-1, 1]
    v2 = path[1:, 1]
    return (v2 - v + 0.0025 * np.cos(3 * p)) / power


def _check_path(path, stored_return, power_scalar, max_steps):
    power = O.mc_power(power_scalar)
    a = _actions_from_path(path, power)
    p2, v2, r, d = O.mc_step(path[:-1, 0], path[:-1, 1], a, power)
    free = (np.abs(path[1:, 1]) < O.MC_MAX_SPEED) & (path[1:, 0] > O.MC_MIN_POSITION) & (np.abs(a) <= 1.0 + 1e-9)
    # positions integrate exactly as the reference recorded them
    assert np.max(np.abs(p2[free] - path[1:, 0][free])) < 1e-12
    # done only at the last transition, exactly when the goal is reached or the limit hits
    n = len(path) - 1
    assert not d[:-1].any()
    assert bool(d[-1]) == (path[-1, 0] >= 0.45)
    assert d[-1] or n == max_steps
    if free.all():
        # return = sum(100*done - 0.1 a^2) -- pins reward, goal threshold and |a|<=1 clipping
        assert abs(r.sum() - stored_return) < 1e-9
        assert np.max(np.abs(a)) <= 1.0 + 1e-9
    return free.all()


def test_reference_summaries_pin_reward_done_and_time_limit(summaries):
    n_exact = 0
    for i in range(int(summaries["n_files"])):
        ps = float(summaries[f"f{i}_power_scalar"])
        ms = int(summaries[f"f{i}_max_steps"])
        n_exact += _check_path(summaries[f"f{i}_best_path"], float(summaries[f"f{i}_best_reward"]), ps, ms)
        for j in range(5):
            n_exact += _check_path(summaries[f"f{i}_last_path{j}"], float(summaries[f"f{i}_last_reward{j}"]), ps, ms)
        eps = summaries[f"f{i}_episodes"]
        assert eps[:, 0].max() == ms            # TimeLimit: 999 stock, 1000 edited env
        assert (eps[:, 0] >= 1).all()
    assert n_exact >= 20


def test_time_limit_semantics():
    d = O.time_limit(np.array([False, False, True]), np.array([998, 999, 5]), 999)
    assert d.tolist() == [False, True, True]


def test_c_oracle_matches_numpy_oracle(oracle_clib):
    rng = np.random.default_rng(1234)
    n = 100000
    pos = rng.uniform(-1.2, 0.6, n)
    vel = rng.uniform(-0.07, 0.07, n)
    act = rng.uniform(-1.5, 1.5, n)
    pos[:4] = [-1.2, -1.2, 0.449, 0.6]
    vel[:4] = [-0.01, 0.0, 0.07, 0.07]
    p2, v2, r, d = O.mc_step(pos, vel, act)
    cp, cv = pos.copy(), vel.copy()
    cr = np.empty(n)
    cd = np.empty(n, np.uint8)
    dp = ctypes.POINTER(ctypes.c_double)
    oracle_clib.ssc_oracle_mc_step(ctypes.c_int64(n), cp.ctypes.data_as(dp), cv.ctypes.data_as(dp),
                                   act.ctypes.data_as(dp), ctypes.c_double(0.0015), cr.ctypes.data_as(dp),
                                   cd.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    assert np.array_equal(cp, p2) and np.array_equal(cv, v2)
    assert np.array_equal(cd.astype(bool), d)
    assert np.max(np.abs(cr - r)) < 1e-13


def test_philox_known_answers(oracle_clib):
    """Random123 kat_vectors for philox4x32-10."""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kats:
        got = O.philox4x32_10(*ctr, *key)
        assert tuple(int(x) for x in got) == exp
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        oracle_clib.ssc_oracle_philox(c, k, o)
        assert tuple(o) == exp


def test_c_rollout_matches_numpy_replay(oracle_clib):
    """The C scalar rollout and the numpy RNG/step restatement agree on a short rollout."""
    n, K, seed, id0, step0 = 257, 23, 1234, 1000, 6
    ids = np.uint64(id0) + np.arange(n, dtype=np.uint64)
    pos0, vel0 = O.mc_reset_state(seed, ids, O.RESET_T0)
    pos, vel = pos0.astype(np.float64), vel0.astype(np.float64)
    steps = np.full(n, 990, np.int32)       # forces a time-limit reset inside the window
    cp, cv, cs = pos.copy(), vel.copy(), steps.copy()
    stats = np.zeros(4)
    dp = ctypes.POINTER(ctypes.c_double)
    oracle_clib.ssc_oracle_mc_rollout_random.restype = ctypes.c_int64
    oracle_clib.ssc_oracle_mc_rollout_random(
        ctypes.c_int64(n), ctypes.c_int32(K), cp.ctypes.data_as(dp), cv.ctypes.data_as(dp),
        cs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_double(0.0015), ctypes.c_int32(999),
        ctypes.c_uint64(seed), ctypes.c_uint64(id0), ctypes.c_uint64(step0), stats.ctypes.data_as(dp))
    el = steps.astype(np.int64)
    tot_r = 0.0
    n_eps = 0
    for k in range(K):
        t = step0 + k
        a = O.random_policy_actions(seed, ids, t).astype(np.float64)
        pos, vel, r, d = O.mc_step(pos, vel, a)
        el += 1
        d = O.time_limit(d, el, 999)
        tot_r += r.sum()
        if d.any():
            rp, rv = O.mc_reset_state(seed, ids, t)
            pos = np.where(d, rp.astype(np.float64), pos)
            vel = np.where(d, 0.0, vel)
            el = np.where(d, 0, el)
            n_eps += int(d.sum())
    assert np.array_equal(cp, pos) and np.array_equal(cv, vel)
    assert np.array_equal(cs, el)
    assert n_eps == n and stats[3] == n_eps
    assert abs(stats[0] - tot_r) < 1e-9


# ------------------------------------------------------------------ geometry helpers --
def test_geometry_against_reference_numerical(kats):
    for c in range(int(kats["n_geom"])):
        radii, a, b, pt = (kats[f"g{c}_{k}"] for k in ("radii", "a", "b", "pt"))
        dist = O.distance_func(radii)
        assert np.allclose(dist(a, b), kats[f"g{c}_dist_ab"], rtol=1e-14, atol=0)
        assert np.allclose(O.projection_of_a_onto_b(a, b), kats[f"g{c}_proj"], rtol=1e-13, atol=1e-15)
        assert np.allclose(O.projection_of_a_onto_b(a, b, radii=radii), kats[f"g{c}_proj_radii"], rtol=1e-13, atol=1e-15)
        got = O.dist_line_seg_to_point(a, b, pt, dist, radii)
        assert np.allclose(got, kats[f"g{c}_segdist"], rtol=1e-12, atol=1e-14)
        if len(a) > 1:
            # the quirk: batched != row-wise
            assert not np.allclose(kats[f"g{c}_segdist"], kats[f"g{c}_segdist_rowwise"])


def test_reference_unit_test_vectors_for_projection():
    """tests/utilities/test_numerical.py:33-58 of the reference (1-D inputs)."""
    assert np.array_equal(O.projection_of_a_onto_b(np.array([1, 1]), np.array([0, 1])), np.array([0, 1]))
    assert np.allclose(O.projection_of_a_onto_b(np.array([1, 1, 1]), np.array([0, 1, 1])), np.array([0, 1, 1]))
    d = O.distance_func([1, 1])
    assert np.isclose(O.dist_line_seg_to_point(np.array([0, 1]), np.array([1, 2]), np.array([2, 1]), d, [1, 1]), 2 ** .5)
    s = 10
    d = O.distance_func([1, s])
    assert np.isclose(O.dist_line_seg_to_point(np.array([0, s]), np.array([1, 2 * s]), np.array([2, s]), d, [1, s]), 2 ** .5)


def test_path_statistics_and_shortcutter(kats):
    for c in range(int(kats["n_paths"])):
        path = kats[f"p{c}_path"]
        stds, means = O.path_deltas_stds_and_means_per_dim(path)
        assert np.allclose(stds, kats[f"p{c}_stds"], rtol=1e-12)
        assert np.allclose(means, kats[f"p{c}_means"], rtol=1e-12)
        radii = O.radii_calc(means, stds, 1, 1, 1)
        assert np.allclose(radii, kats[f"p{c}_radii"], rtol=1e-12)
        short = O.path_shortcutter(path, O.distance_func(kats[f"p{c}_radii"]), 1)
        assert short.shape == kats[f"p{c}_short"].shape
        assert np.array_equal(short, kats[f"p{c}_short"])
    # reference tests/utilities/test_numerical.py:88-103
    assert O.length_weighted_activities([[1, 4], [2, 8], [3, 11], [5, 7], [8, 15], [13, 18]], sub_extra=0)[0] == 13
    d = O.distance_func([1, 1])
    assert np.array_equal(O.path_shortcutter([[0, 0], [1, 1], [2, 2], [3, 3], [1, 1]], d, 1), [[0, 0], [1, 1], [1, 1]])
    p = [[0, 0], [1, 1], [2, 2], [3, 3], [4, 4]]
    assert np.array_equal(O.path_shortcutter(p, d, 1), p)


def test_activity_solver(kats):
    for c in range(int(kats["n_act"])):
        w, chosen = O.length_weighted_activities(kats[f"act{c}_in"].tolist())
        assert w == int(kats[f"act{c}_w"])
        assert np.array_equal(np.asarray(chosen, np.int64).reshape(-1, 2), kats[f"act{c}_chosen"])


def test_mpc_scores_against_reference_helpers(kats):
    for c in range(int(kats["n_mpc"])):
        S, wp, left, radii = (kats[f"m{c}_{k}"] for k in ("S", "wp", "left", "radii"))
        scores, best_score, best, idx = O.mpc_scores_add_delta(S, wp, left, radii, int(kats[f"m{c}_cur"]))
        assert np.allclose(scores, kats[f"m{c}_scores"], rtol=1e-11, atol=1e-12)
        assert np.array_equal(idx, kats[f"m{c}_final_idx"])
        assert best == int(kats[f"m{c}_best"])
        # distances_left restatement
        assert np.allclose(O.distances_left(wp, O.distance_func(radii)), left, rtol=1e-13)
