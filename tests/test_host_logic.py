"""Host-side product logic on the CPU (no GPU): the navigator's planning helpers and the replay
buffer against KATs generated from the reference's own numerical.py / replay_buffer.py, and
against the oracle."""
import numpy as np
import pytest

from oracle import ssc_oracle as O
from smartstartcontinuous_amd import numerical as N
from smartstartcontinuous_amd.replay_buffer import ReplayBuffer


@pytest.fixture(scope="module")
def kats(golden_dir):
    return np.load(f"{golden_dir}/numerical_kats.npz")


def test_path_statistics_radii_shortcutter(kats):
    for c in range(int(kats["n_paths"])):
        path = kats[f"p{c}_path"]
        stds, means = N.path_deltas_stds_and_means_per_dim(path)
        assert np.allclose(stds, kats[f"p{c}_stds"], rtol=1e-12) and np.allclose(means, kats[f"p{c}_means"], rtol=1e-12)
        radii = N.radii_calc(means, stds, 1, 1, 1)
        assert np.allclose(radii, kats[f"p{c}_radii"], rtol=1e-12)
        dist = N.elliptical_euclidean_distance_function_generator(kats[f"p{c}_radii"])
        assert np.array_equal(N.path_shortcutter(path, dist, 1), kats[f"p{c}_short"])                   # libssc's host routine
        assert np.array_equal(N.path_shortcutter(path, dist, 1, native=False), kats[f"p{c}_short"])     # the numpy statement
    # the reference's own unit-test vectors (tests/utilities/test_numerical.py:10-31, 88-103)
    assert N.path_deltas_stds_and_means_per_dim([[1], [2], [4]])[0][0] == .50
    assert N.length_weighted_activities_solver([[1, 4], [2, 8], [3, 11], [5, 7], [8, 15], [13, 18]])[0] == 13
    d = N.elliptical_euclidean_distance_function_generator([1, 1])
    assert np.array_equal(N.path_shortcutter([[0, 0], [1, 1], [2, 2], [3, 3], [1, 1]], d, 1), [[0, 0], [1, 1], [1, 1]])
    p = [[0, 0], [1, 1], [2, 2], [3, 3], [4, 4]]
    assert np.array_equal(N.path_shortcutter(p, d, 1), p)


def test_activity_solver_kats(kats):
    for c in range(int(kats["n_act"])):
        w, chosen = N.length_weighted_activities_solver(kats[f"act{c}_in"].tolist(), sub_extra=1)
        assert w == int(kats[f"act{c}_w"])
        assert np.array_equal(np.asarray(chosen, np.int64).reshape(-1, 2), kats[f"act{c}_chosen"])


def test_distance_volume_distances_left(kats):
    for c in range(int(kats["n_geom"])):
        radii, a, b = kats[f"g{c}_radii"], kats[f"g{c}_a"], kats[f"g{c}_b"]
        assert np.allclose(N.elliptical_euclidean_distance_function_generator(radii)(a, b), kats[f"g{c}_dist_ab"], rtol=1e-14)
    assert np.allclose([N.volume_of_n_dimensional_hyperellipsoid(list(r[:2])) for r in kats["vol_radii"]], kats["vol_2d"])
    assert np.allclose([N.volume_of_n_dimensional_hyperellipsoid(list(r)) for r in kats["vol_radii"]], kats["vol_3d"])
    for c in range(int(kats["n_mpc"])):
        wp, radii = kats[f"m{c}_wp"], kats[f"m{c}_radii"]
        left = N.distances_left(wp, N.elliptical_euclidean_distance_function_generator(radii))
        assert np.allclose(left, kats[f"m{c}_left"], rtol=1e-12)
    assert np.array_equal(N.get_start_waypoints_final_states_steps(np.arange(10)[:, None], 3)[:, 0], [0, 3, 6, 9])
    assert np.array_equal(N.get_start_waypoints_final_states_steps(np.arange(11)[:, None], 3)[:, 0], [0, 3, 6, 9, 10])


def test_replay_buffer_matches_reference_trace(golden_dir):
    """Same add / start_new_episode sequence as tests/golden/make_goldens.py:replay_buffer_kats ran
    through the REFERENCE ReplayBuffer: identical bookkeeping after every add."""
    g = np.load(f"{golden_dir}/replay_buffer_kats.npz")
    agent = object()
    buf = ReplayBuffer(agent, 50)
    k, i = 0, 0
    for L in g["ep_lens"]:
        buf.start_new_episode(agent)
        for _ in range(int(L)):
            buf.add(agent, np.array([k, 0.0]), np.array([0.0]), 0.0, False, np.array([k + 1, 0.0]))
            buf.add(object(), np.array([-1, 0.0]), np.array([0.0]), 0.0, False, np.array([-1, 0.0]))  # ignored writer
            k += 1
            assert len(buf.buffer) == g["trace_len"][i] and buf.next_episode_number == g["trace_next"][i]
            starts = list(buf.episode_starting_indices)
            assert starts == [v for v in g["trace_starts"][i] if v != -1][:len(starts)] and len(starts) == int((g["trace_starts"][i] != -1).sum())
            i += 1
    assert buf.episode_number_to_buffer_index(buf.episode_starting_indices[0]) == int(g["final_first_index"])
    for j in range(3):
        path = np.asarray(buf.get_episodic_path_to_buffer_index(int(g[f"path_idx_{j}"])))
        assert np.array_equal(path, g[f"path_to_{j}"])
    assert np.array_equal(buf.get_all_states(), g["all_states"])
    s, a, r, t, s2 = buf.sample_batch(16)
    assert s.shape == (16, 2) and a.shape == (16, 1) and len(set(s[:, 0].tolist())) == 16
    idx = buf.get_possible_smart_start_indices(10)
    assert len(idx) == 10 and idx.min() >= int(g["final_first_index"])
    # reference tests/RLAgents/test_replayBuffer.py semantics: FIFO eviction keeps the newest max_buffer_size
    assert len(buf) == 50 and buf.buffer[-1][0][0] == k - 1 and buf.buffer[0][0][0] == k - 50


def test_replay_buffer_clear_and_counter_assignment_match_reference_trace(golden_dir):
    """clear() and ``next_episode_number = v`` (replay_buffer.py:105-107, :131): the reference moves the counter and keeps
    the VALUES of episode_starting_indices; tests/golden/replay_buffer_clear_kats.npz holds its (len, counter, starts)
    after every operation of the script make_goldens.py:replay_buffer_clear_kats ran through it."""
    g = np.load(f"{golden_dir}/replay_buffer_clear_kats.npz")
    agent = object()
    buf = ReplayBuffer(agent, 40)
    k, i = 0, 0

    def check(op):
        nonlocal i
        row = g["trace"][i]
        assert int(g["ops"][i]) == op
        starts = [int(v) for v in row[2:] if v != -99]
        assert (len(buf.buffer), buf.next_episode_number, list(buf.episode_starting_indices)) == (int(row[0]), int(row[1]), starts), (i, op)
        i += 1
    for kind, arg in zip(g["script_kind"], g["script_arg"]):
        if kind == 0:
            buf.start_new_episode(agent)
            check(0)
            for _ in range(int(arg)):
                buf.add(agent, np.array([k, 0.0]), np.array([0.0]), 0.0, False, np.array([k + 1, 0.0]))
                k += 1
                check(1)
        elif kind == 1:
            buf.clear()
            check(2)
        else:
            buf.next_episode_number = int(arg)
            check(3)
    assert i == len(g["ops"]) == 74


def test_path_shortcutter_takes_the_numpy_route_at_eight_dimensions():
    """d == 8 is where np.sum over the last axis switches to its 8-accumulator pairwise unrolling: the native routine (strict
    index order) is used for d < 8 only, so both routes give the numpy decisions -- checked with pairs placed within an
    ulp of theta."""
    from smartstartcontinuous_amd import numerical as num
    rng = np.random.default_rng(3)
    for d in (3, 7, 8):
        radii = rng.uniform(0.5, 2.0, d)
        dist = num.elliptical_euclidean_distance_function_generator(radii)
        path = np.cumsum(rng.normal(size=(60, d)) * 0.3, axis=0)
        path[20] = path[5] + radii * np.sqrt(1.0 / d) * (1.0 - 1e-16)        # a pair at distance theta = 1 to within an ulp
        a = num.path_shortcutter(path, dist, 1.0)
        b = num.path_shortcutter(path, dist, 1.0, native=False)
        assert np.array_equal(a, b), d


def test_rltrain_loop_with_fake_env_and_agent():
    """rlTrain control flow (rlTrain.py:63-114) with CPU stand-ins: break on done, max_steps cap,
    agent call order, Summary records."""
    from smartstartcontinuous_amd.rl_train import rlTrain
    from smartstartcontinuous_amd.agents import RLAgent

    class Env:
        class spec:
            id = "Fake-v0"

        def __init__(self):
            self.t = 0

        def reset(self):
            self.t = 0
            return np.zeros(2)

        def step(self, a):
            self.t += 1
            return np.full(2, self.t), -1.0, self.t >= 7, {}

    calls = []

    class Agent(RLAgent):
        def get_action(self, s):
            calls.append("act")
            return np.zeros(1)

        def observe(self, *a):
            calls.append("obs")

        def render(self, env, **kw):
            return False

        def start_new_episode(self, s):
            calls.append("start")

        def end_episode(self):
            calls.append("end")

    summ = rlTrain(Agent(), Env(), print_results=False, print_steps=False, num_episodes=3, max_steps=5)
    assert summ.episodes == [(5, -5.0)] * 3 and summ.name == "Agent_Fake-v0"
    summ = rlTrain(Agent(), Env(), print_results=False, print_steps=False, num_episodes=2, max_steps=100)
    assert summ.episodes == [(7, -7.0)] * 2 and len(summ.best_path) == 8
    assert calls[:4] == ["start", "act", "obs", "act"] and calls.count("end") == 5


def test_native_path_shortcut_makes_the_numpy_decisions():
    """ssc_path_shortcut (csrc/path_geometry.cpp, host code) against the numpy statement of numerical.py:189-246 and against the
    oracle: random walks in 1-4 dimensions, paths on an integer grid (exact ties in distance AND in shortcut weight, where
    the reference's `inc >= best` rule decides), paths that revisit states, degenerate lengths, several thetas."""
    rng = np.random.default_rng(42)
    cases = []
    for n in (0, 1, 2, 3, 4, 7, 40, 120, 300):
        for d in (1, 2, 3, 4):
            cases.append((np.cumsum(rng.normal(0, 0.01, (n, d)), axis=0), None))
    for n in (6, 25, 90):
        for d in (1, 2):
            grid = np.cumsum(rng.integers(-1, 2, (n, d)), axis=0).astype(np.float64)      # integer lattice walk
            cases.append((grid, np.ones(d)))
    loop = np.concatenate([np.linspace([0, 0], [1, 1], 30), np.linspace([1, 1], [0, 0], 30)[1:]])
    cases.append((loop, np.array([0.05, 0.05])))
    checked = 0
    for path, radii in cases:
        if radii is None:
            if len(path) < 2:
                radii = np.ones(path.shape[1])
            else:
                stds, means = N.path_deltas_stds_and_means_per_dim(path)
                radii = N.radii_calc(means, stds, 1, 1, 1)
        dist = N.elliptical_euclidean_distance_function_generator(radii)
        for theta in (0.5, 1.0, 2.0):
            a = N.path_shortcutter(path, dist, theta)
            if len(path) >= 1:
                b = N.path_shortcutter(path, dist, theta, native=False)
                assert a.shape == b.shape and np.array_equal(a, b), (path.shape, theta)
                if len(path) >= 2:
                    assert np.array_equal(a, O.path_shortcutter(path, O.distance_func(radii), theta))
                assert len(a) <= len(path) and (len(path) == 0 or (np.array_equal(a[0], path[0]) and np.array_equal(a[-1], path[-1])))
                checked += 1
            else:
                assert a.shape == (0, path.shape[1])
    assert checked >= 100
    # a generic distance function (no radii attribute) still runs the numpy statement
    plain = lambda x, y: np.sqrt(np.sum((np.asarray(x) - np.asarray(y)) ** 2, axis=-1))
    p = np.cumsum(rng.normal(0, 0.3, (30, 2)), axis=0)
    d1 = N.elliptical_euclidean_distance_function_generator([1.0, 1.0])
    assert np.array_equal(N.path_shortcutter(p, plain, 1.0), N.path_shortcutter(p, d1, 1.0))
    # argument checks of the C entry point
    from smartstartcontinuous_amd import _ffi
    lib = _ffi.lib()
    assert lib.ssc_path_shortcut(None, 5, 2, None, 1.0, None, None) == _ffi.SSC_EINVAL
    assert lib.ssc_path_shortcut(None, 5, 9, None, 1.0, None, None) == _ffi.SSC_EINVAL and b"ssc_path_shortcut" in lib.ssc_last_error()
    r = np.array([1.0, 0.0])
    k = np.ones(3, np.uint8)
    pth = np.zeros((3, 2))
    import ctypes
    assert lib.ssc_path_shortcut(pth.ctypes.data_as(ctypes.c_void_p), 3, 2, r.ctypes.data_as(ctypes.c_void_p), 1.0,
                                 k.ctypes.data_as(ctypes.c_void_p), None) == _ffi.SSC_EINVAL          # radii must be positive


def test_oracle_and_product_helpers_agree():
    rng = np.random.default_rng(0)
    path = np.cumsum(rng.normal(size=(60, 2)) * [0.02, 0.004], axis=0)
    path = np.concatenate([path, path[::-1][:20] + 1e-3])
    stds, means = N.path_deltas_stds_and_means_per_dim(path)
    so, mo = O.path_deltas_stds_and_means_per_dim(path)
    assert np.allclose(stds, so) and np.allclose(means, mo)
    radii = N.radii_calc(means, stds, 1, 1, 1)
    assert np.array_equal(N.path_shortcutter(path, N.elliptical_euclidean_distance_function_generator(radii), 1),
                          O.path_shortcutter(path, O.distance_func(radii), 1))


def test_summary_reads_and_writes_the_reference_json_schema(golden_dir, tmp_path):
    """Summary.load on one of the reference's own dumps (tests/golden/reference_summary.json), and the
    files we write have exactly the same keys (datacontainers.py:257-374)."""
    import json
    import sys
    from smartstartcontinuous_amd.rl_train import Episode, Summary
    ref = Summary.load(f"{golden_dir}/reference_summary.json")
    raw = json.load(open(f"{golden_dir}/reference_summary.json"))
    assert ref.name.startswith("SmartStartC_DDPG_Baselines_agent_MountainCarContinuous-v0")
    assert len(ref) == 60 and ref.steps_episode()[0] == raw["episodes"][0][0]
    assert ref.get_best_path_and_reward()[1] == raw["best_reward"] and len(ref.last_paths) == raw["last_x"] == 5
    assert ref.param_dict["n_ss"] == 2000 and ref.name_of_agent == "SmartStartContinuous"
    assert abs(ref.total_reward() - sum(r for _, r in raw["episodes"])) < 1e-9
    # our own summaries: same schema, save() auto-increments the postfix like the reference
    s = Summary("Agent_Env-v0")
    assert json.loads(s.to_json())["best_reward"] == -sys.maxsize
    for k in range(7):
        ep = Episode()
        for t in range(3 + k):
            ep.append(np.array([t, 0.0]), np.array([0.5]), -1.0 + k, np.array([t + 1, 0.0]), t == 2 + k)
        s.append(ep)
    s.start_smart_start_episode()
    f0 = s.save(str(tmp_path))
    f1 = s.save(str(tmp_path))
    assert f0.endswith("Agent_Env-v0_0.json") and f1.endswith("Agent_Env-v0_1.json")
    back = Summary.load(f0)
    assert set(json.load(open(f0)).keys()) == set(raw.keys())
    assert [tuple(e) for e in back.episodes] == s.episodes and back.best_reward == s.best_reward == (3 + 6) * 5.0
    assert len(back.last_paths) == 5 and len(back.best_path) == 3 + 6 + 1 and back.smart_start_episodes == [7]


def test_oracle_replay_ring_and_sampler():
    """The restated FIFO ring equals the reference-pinned host ReplayBuffer's content order, and the sampler
    returns distinct in-range indices (random.sample semantics, replay_buffer.py:79-83)."""
    from oracle import ssc_oracle as O
    rng = np.random.default_rng(0)
    cap, n = 37, 100
    ring = dict(s=np.zeros((cap, 2), np.float32), a=np.zeros((cap, 1), np.float32), r=np.zeros(cap, np.float32),
                t=np.zeros(cap, np.uint8), s2=np.zeros((cap, 2), np.float32))
    s, s2 = rng.normal(size=(n, 2)).astype(np.float32), rng.normal(size=(n, 2)).astype(np.float32)
    a, r = rng.normal(size=(n, 1)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    t = (rng.random(n) < 0.1).astype(np.uint8)
    count = O.replay_append(ring, 0, s[:60], a[:60], r[:60], t[:60], s2[:60], reward_scale=0.5)
    count = O.replay_append(ring, count, s[60:], a[60:], r[60:], t[60:], s2[60:], reward_scale=0.5)
    assert count == n
    # the ring holds exactly the newest `cap` records, record j at row j % cap
    for j in range(n - cap, n):
        assert np.array_equal(ring["s"][j % cap], s[j]) and ring["r"][j % cap] == np.float32(r[j]) * np.float32(0.5)
        assert ring["t"][j % cap] == t[j] and np.array_equal(ring["s2"][j % cap], s2[j])
    # the host ReplayBuffer (pinned by the reference's own tests) keeps the same records, oldest first
    from smartstartcontinuous_amd.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(None, cap)
    for j in range(n):
        rb.add(None, s[j], a[j], float(np.float32(r[j]) * np.float32(0.5)), bool(t[j]), s2[j])
    hs, ha, hr, ht, hs2 = rb.all_batch()
    order = [(j % cap) for j in range(n - cap, n)]
    assert np.allclose(hs, ring["s"][order]) and np.allclose(hr, ring["r"][order]) and np.allclose(hs2, ring["s2"][order])
    # sampler: distinct, in range, reproducible, and the degenerate size == batch case is a permutation
    idx = O.replay_sample_indices(7, 3, 1000, 5, 64)
    assert idx.shape == (5, 64) and idx.min() >= 0 and idx.max() < 1000
    assert all(len(set(row.tolist())) == 64 for row in idx)
    assert np.array_equal(idx, O.replay_sample_indices(7, 3, 1000, 5, 64))
    assert not np.array_equal(idx[0], idx[1])
    full = O.replay_sample_indices(1, 0, 64, 2, 64)
    assert all(sorted(row.tolist()) == list(range(64)) for row in full)
    # batches above 64 (slot in 16 counter bits): distinct, dense draws included; batch <= 64 streams are unchanged by it
    big = O.replay_sample_indices(7, 3, 700, 2, 512)
    assert big.shape == (2, 512) and all(len(set(row.tolist())) == 512 for row in big) and big.max() < 700
    assert sorted(O.replay_sample_indices(2, 0, 300, 1, 300)[0].tolist()) == list(range(300))


def test_add_noise_matches_reference_rule():
    """helper_funcs.add_noise (NN_Dynamics_Model/helper_funcs.py:10-17): noise std = column mean * ratio, and ONLY
    for columns whose mean is positive."""
    from smartstartcontinuous_amd.agents import add_noise
    rng = np.random.RandomState(0)
    data = np.stack([np.full(20000, 2.0), np.full(20000, -3.0), np.zeros(20000), np.linspace(0.0, 1.0, 20000)], axis=1)
    out = add_noise(data, 0.01, rng)
    assert out is not data and np.array_equal(out[:, 1], data[:, 1]) and np.array_equal(out[:, 2], data[:, 2])
    assert abs((out[:, 0] - data[:, 0]).std() - 0.02) < 1e-3 and abs((out[:, 0] - data[:, 0]).mean()) < 1e-3
    assert abs((out[:, 3] - data[:, 3]).std() - 0.005) < 3e-4
    assert add_noise(np.zeros((0, 3)), 0.01).shape == (0, 3)


def test_oracle_episode_index_matches_reference_trace(golden_dir):
    """The restatement of the VECTORISED ring's episode index (oracle.replay_episode_steps / smart_start_valid /
    replay_episode_path -- what the device kernels are checked against) on the reference buffer's own trace
    (replay_buffer_kats.npz, n = 1): same first smart-start index and the same episodic paths as
    smartstart/RLAgents/replay_buffer.py:136-176 produced."""
    from oracle import ssc_oracle as O
    g = np.load(f"{golden_dir}/replay_buffer_kats.npz")
    total, cap = int(g["ep_lens"].sum()), 50
    done = np.zeros((total, 1), bool)
    done[np.cumsum(g["ep_lens"]) - 1, 0] = True
    steps, run = O.replay_episode_steps(done)
    assert run[0] == 0 and steps[:, 0].max() == g["ep_lens"].max()
    s = np.stack([np.arange(total, dtype=float), np.zeros(total)], 1)
    s_ring, s2_ring, ring_steps = np.zeros((cap, 2)), np.zeros((cap, 2)), np.zeros(cap, np.int64)
    for j in range(total):                       # FIFO: record j lives at j % capacity
        s_ring[j % cap], s2_ring[j % cap], ring_steps[j % cap] = s[j], s[j] + [1, 0], steps[j, 0]
    valid = O.smart_start_valid(ring_steps, cap, total, 1)
    first = int(g["final_first_index"])
    assert not valid[:first].any() and valid[first:].all()
    for j in range(3):
        path = O.replay_episode_path(s_ring, s2_ring, ring_steps, cap, total, 1, int(g[f"path_idx_{j}"]), 1000)
        assert np.array_equal(path, g[f"path_to_{j}"])
    idx = O.smart_start_indices(valid, 20, 5, 0)
    assert (idx >= first).all() and len(set(idx.tolist())) == 20
    more = O.smart_start_indices(valid, 64, 5, 1)            # more slots than valid records: every valid index once
    assert sorted(more[more >= 0].tolist()) == np.nonzero(valid)[0].tolist()


def test_replay_buffer_save_load_roundtrip(tmp_path):
    """ReplayBuffer.save / load (replay_buffer.py:117-134): the pickled record list round-trips; episode markers are
    not persisted (the reference sets next_episode_number = len(buffer) and keeps no starts)."""
    agent = object()
    a = ReplayBuffer(agent, 20)
    a.start_new_episode(agent)
    for k in range(27):
        a.add(agent, np.array([k, -k], float), np.array([0.5 * k]), float(k), k % 5 == 4, np.array([k + 1, -k - 1], float))
    path = str(tmp_path / "replay_buffer.obj")
    a.save(path)
    b = ReplayBuffer(agent, 20)
    assert b.load(str(tmp_path / "missing.obj")) is False and len(b) == 0
    assert b.load(path) is True and len(b) == 20 and b.next_episode_number == 20 and len(b.episode_starting_indices) == 0
    for i in range(20):
        for x, y in zip(a.buffer[i], b.buffer[i]):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    small = ReplayBuffer(agent, 8)          # a smaller buffer keeps the newest records
    assert small.load(path) and len(small) == 8 and small.buffer[-1][0][0] == 26 and small.buffer[0][0][0] == 19


def test_vectorised_activity_solver_equals_the_reference_loop():
    """length_weighted_activities_solver fills the reference's table one end time at a time (one vectorised max per row);
    on random interval sets -- unsorted and in np.where order, with and without sub_extra -- it returns exactly what the
    interval-by-interval loop of numerical.py:189-222 returns (weight AND chosen intervals, i.e. the same tie-breaking)."""
    from smartstartcontinuous_amd import numerical as N
    rng = np.random.default_rng(0)
    for trial in range(120):
        L, n = int(rng.integers(10, 120)), int(rng.integers(65, 700))
        s = rng.integers(0, L - 2, n)
        e = s + rng.integers(1, L - s)
        acts = np.stack([s, e], 1)
        if trial % 3 == 0:
            acts = acts[np.lexsort((acts[:, 1], acts[:, 0]))]
        for sub_extra in (0, 1):
            assert N.length_weighted_activities_solver(acts, sub_extra) == N._length_weighted_activities_loop(acts.tolist(), sub_extra)
