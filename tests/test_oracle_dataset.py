"""Data collection -> dynamics training set on the CPU: the oracle's restatement of
generate_training_data_inputs / _outputs against vectors produced by the REFERENCE's own functions
(tests/golden/data_manipulation_kats.npz, made by importing data_manipulation.py), the host mirror in
smartstartcontinuous_amd.collect_samples against the same vectors, and the statistics / noise helpers."""
import numpy as np
import pytest

from oracle import ssc_oracle as O


@pytest.fixture(scope="module")
def kats(golden_dir):
    return np.load(f"{golden_dir}/data_manipulation_kats.npz")


def split(flat, lens):
    cuts = np.cumsum(lens)[:-1]
    return np.split(flat, cuts, axis=0)


@pytest.mark.parametrize("case", [0, 1])
def test_generate_training_data_matches_reference(kats, case):
    lens = kats[f"c{case}_lens"]
    states, controls = split(kats[f"c{case}_states"], lens), split(kats[f"c{case}_controls"], lens)
    X, Y = O.generate_training_data_inputs(states, controls)
    Z = O.generate_training_data_outputs(states)
    assert np.array_equal(X, kats[f"c{case}_dataX"]) and np.array_equal(Y, kats[f"c{case}_dataY"])
    assert np.array_equal(Z, kats[f"c{case}_dataZ"])
    assert len(X) == int(np.sum(np.maximum(lens - 1, 0)))      # every rollout loses its last entry
    # the host mirror a reference user would import
    from smartstartcontinuous_amd import collect_samples as cs
    X2, Y2 = cs.generate_training_data_inputs(states, controls)
    assert np.array_equal(X2, X) and np.array_equal(Y2, Y)
    assert np.array_equal(cs.generate_training_data_outputs(states), Z)


def chunk_from_rollouts(states, controls, K, rng):
    """A [K, n] transition chunk whose env i replays rollout i, ends it with a terminal flag (unless it fills the
    chunk) and then carries on with unrelated data -- what an auto-resetting rollout leaves behind."""
    n, d = len(states), states[0].shape[1]
    obs = rng.normal(size=(K, n, d))
    act = rng.normal(size=(K, n, 1))
    done = (rng.random((K, n)) < 0.05).astype(np.uint8)
    for i, (s, c) in enumerate(zip(states, controls)):
        L = len(s)
        obs[:L, i], act[:L, i] = s, c
        done[:L, i] = 0
        if L < K:
            done[L - 1, i] = 1
    return obs, act, done


@pytest.mark.parametrize("case", [0, 1])
def test_rollouts_from_chunk_then_format_matches_reference(kats, case):
    """collect_samples semantics (stop after the first terminal step) + formatting, against the reference vectors."""
    lens = kats[f"c{case}_lens"]
    states, controls = split(kats[f"c{case}_states"], lens), split(kats[f"c{case}_controls"], lens)
    obs, act, done = chunk_from_rollouts(states, controls, 333, np.random.default_rng(case))
    st, ct = O.rollouts_from_chunk(obs, act, done)
    assert [len(s) for s in st] == lens.tolist()
    X, Y = O.generate_training_data_inputs(st, ct)
    assert np.array_equal(X, kats[f"c{case}_dataX"]) and np.array_equal(Y, kats[f"c{case}_dataY"])
    assert np.array_equal(O.generate_training_data_outputs(st), kats[f"c{case}_dataZ"])


def test_reference_validation_rollouts_format_to_step_deltas(golden_dir):
    """The reference's recorded validation rollouts, formatted, are exactly the env's own step deltas."""
    r = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    S, A = r["states_val"], r["controls_val"]
    X, Y = O.generate_training_data_inputs(list(S), list(A))
    Z = O.generate_training_data_outputs(list(S))
    assert X.shape == (20 * 332, 2) and Y.shape == (20 * 332, 1)
    p2, v2, _, _ = O.mc_step(X[:, 0], X[:, 1], Y[:, 0])
    assert np.max(np.abs(p2 - X[:, 0] - Z[:, 0])) < 1e-15 and np.max(np.abs(v2 - X[:, 1] - Z[:, 1])) < 1e-15


def test_column_stats_and_zscore():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(1000, 4)) * [1.0, 1e-3, 50.0, 0.0] + [0.5, -2.0, 3.0, 7.0]
    mean, std = O.column_stats(x)
    assert np.allclose(mean, x.mean(0), rtol=0, atol=0) and np.allclose(std, x.std(0), rtol=1e-14)
    z = O.zscore(x, mean, std)
    assert z.dtype == np.float32
    assert np.allclose(z[:, :3].mean(0), 0, atol=1e-6) and np.allclose(z[:, :3].std(0), 1, atol=1e-5)
    assert np.all(z[:, 3] == 0)                                   # 0/0 -> nan -> 0 (np.nan_to_num)
    z2 = O.zscore(np.array([[1.0], [3.0]]), np.array([2.0]), np.array([0.0]))
    assert z2[0, 0] == -np.finfo(np.float32).max and z2[1, 0] == np.finfo(np.float32).max


def test_add_noise_keyed_follows_the_reference_rule():
    """helper_funcs.py:10-17: noise only where mean * noiseToSignal > 0, std = |mean| * noiseToSignal."""
    rng = np.random.default_rng(6)
    x = (rng.normal(size=(20000, 3)) * 0.1 + [2.0, -2.0, 0.0]).astype(np.float32)
    x[:, 2] -= x[:, 2].mean()
    mean = x.astype(np.float64).mean(0)
    mean[2] = 0.0
    y = O.add_noise_keyed(x, mean, 0.01, seed=1234, stream_id=3)
    assert np.array_equal(y[:, 1], x[:, 1]) and np.array_equal(y[:, 2], x[:, 2])
    delta = (y[:, 0] - x[:, 0]).astype(np.float64)
    assert abs(delta.std() / (0.01 * mean[0]) - 1.0) < 0.02 and abs(delta.mean()) < 3 * 0.02 / np.sqrt(20000)
    assert not np.array_equal(O.add_noise_keyed(x, mean, 0.01, 1234, 4)[:, 0], y[:, 0])    # streams differ
    assert np.array_equal(O.add_noise_keyed(x, mean, 0.01, 1234, 3), y)                    # keyed => repeatable
