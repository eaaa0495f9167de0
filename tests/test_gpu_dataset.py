"""GPU parity of the data-collection path (csrc/dataset.hip through the C ABI): rollout chunk -> (dataX, dataY,
dataZ) against the oracle and the reference-generated vectors, column statistics, z-scoring, keyed noise,
CollectSamples end to end and the NND_MB_agent constructor that collects its own training data."""
import numpy as np
import pytest

from oracle import ssc_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ssc():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    import smartstartcontinuous_amd as pkg
    pkg._ffi.lib()
    return pkg


def fill_chunk(ssc, obs, act, obs2, done, packed):
    K, n, d = obs.shape
    ch = ssc.TransitionChunk(d, K, n, "cuda", packed=packed)
    ch.obs.copy_(torch.as_tensor(obs.transpose(2, 0, 1), dtype=torch.float32))
    ch.obs2.copy_(torch.as_tensor(obs2.transpose(2, 0, 1), dtype=torch.float32))
    ch.act.copy_(torch.as_tensor(act[:, :, 0], dtype=torch.float32))
    ch.rew.zero_()
    ch.done.copy_(torch.as_tensor(done))
    return ch


def synthetic_chunk(K, n, d, seed, p_done=0.02):
    rng = np.random.default_rng(seed)
    obs = rng.normal(size=(K, n, d)).astype(np.float32)
    act = rng.uniform(-2, 2, size=(K, n, 1)).astype(np.float32)
    done = (rng.random((K, n)) < p_done).astype(np.uint8)
    if n > 3 and K > 1:
        done[0, 1] = 1            # rollout of length 1: contributes nothing
        done[:, 2] = 0            # rollout that fills the chunk
    obs2 = rng.normal(size=(K, n, d)).astype(np.float32)
    # inside an episode the next observation IS the next step's observation (no reset in between)
    cont = done[:-1] == 0
    obs2[:-1][cont] = obs[1:][cont]
    return obs, act, obs2, done


@pytest.mark.parametrize("K,n,d,packed,p_done", [(333, 25, 2, True, 0.004), (65, 64, 3, False, 0.05),
                                                   (130, 200, 3, True, 0.02), (2, 70, 2, True, 0.3),
                                                   (1, 5, 2, False, 0.0), (40, 1, 3, True, 0.0)])
def test_dataset_from_chunk_matches_oracle(ssc, K, n, d, packed, p_done):
    """Ragged rollouts (a done at step 0 gives an empty rollout, no done gives K-1 rows), env counts and step
    counts off the 64 x 64 tile grid, dense and packed chunk layouts: bit-exact rows, lengths and offsets."""
    from smartstartcontinuous_amd import collect_samples as cs
    obs, act, obs2, done = synthetic_chunk(K, n, d, seed=K * 1000 + n, p_done=p_done)
    ch = fill_chunk(ssc, obs, act, obs2, done, packed)
    ts = cs.dataset_from_chunk(ch)
    st, ct = O.rollouts_from_chunk(obs, act, done)
    lens = np.array([len(s) for s in st])
    assert np.array_equal(ts.lens.cpu().numpy(), lens)
    assert np.array_equal(ts.offsets.cpu().numpy(), np.concatenate([[0], np.cumsum(np.maximum(lens - 1, 0))]))
    X, Y = O.generate_training_data_inputs(st, ct)
    Z = O.generate_training_data_outputs(st)                       # fp32 inputs -> fp32 subtraction, like the kernel
    assert ts.dataX.shape == X.shape and ts.dataY.shape == Y.shape
    assert np.array_equal(ts.dataX.cpu().numpy(), X) and np.array_equal(ts.dataY.cpu().numpy(), Y)
    assert np.array_equal(ts.dataZ.cpu().numpy(), Z)


@pytest.mark.parametrize("case", [0, 1])
def test_dataset_from_chunk_reproduces_reference_vectors(ssc, golden_dir, case):
    """The reference's own generate_training_data_* outputs (data_manipulation_kats.npz) through the device path."""
    from smartstartcontinuous_amd import collect_samples as cs
    k = np.load(f"{golden_dir}/data_manipulation_kats.npz")
    lens = k[f"c{case}_lens"]
    cuts = np.cumsum(lens)[:-1]
    states, controls = np.split(k[f"c{case}_states"], cuts), np.split(k[f"c{case}_controls"], cuts)
    K, n, d = 333, len(lens), states[0].shape[1]
    rng = np.random.default_rng(case)
    obs = rng.normal(size=(K, n, d)); act = rng.normal(size=(K, n, 1)); obs2 = rng.normal(size=(K, n, d))
    done = np.zeros((K, n), np.uint8)
    for i, (s, c) in enumerate(zip(states, controls)):
        L = len(s)
        obs[:L, i], act[:L, i] = s, c
        obs2[:L - 1, i] = s[1:]
        if L < K:
            done[L - 1, i] = 1
    ts = cs.dataset_from_chunk(fill_chunk(ssc, obs, act, obs2, done, packed=True))
    f32 = lambda a: a.astype(np.float32)
    assert np.array_equal(ts.dataX.cpu().numpy(), f32(k[f"c{case}_dataX"]))
    assert np.array_equal(ts.dataY.cpu().numpy(), f32(k[f"c{case}_dataY"]))
    # the states are stored in fp32, so the delta carries their rounding: |err| <= 2 ulp of the states
    scale = np.abs(k[f"c{case}_states"]).max()
    assert np.max(np.abs(ts.dataZ.cpu().numpy() - k[f"c{case}_dataZ"])) <= 2.4e-7 * scale


def test_column_stats_zscore_and_noise(ssc):
    from smartstartcontinuous_amd import collect_samples as cs
    rng = np.random.default_rng(11)
    x = (rng.normal(size=(100003, 4)) * [1.0, 1e-3, 50.0, 0.0] + [0.5, -2.0, 3.0, 7.0]).astype(np.float32)
    xd = torch.as_tensor(x, device="cuda")
    mean, std = cs.column_stats(xd)
    m_ref, s_ref = O.column_stats(x)
    assert np.allclose(mean.cpu().numpy(), m_ref, rtol=1e-13, atol=1e-15)
    assert np.allclose(std.cpu().numpy(), s_ref, rtol=1e-12, atol=1e-15) and std[3].item() == 0.0
    mean2, std2 = cs.column_stats(xd)
    assert torch.equal(mean, mean2) and torch.equal(std, std2)          # fixed summation order
    # z-score into the network-input matrix: columns 1..4 of a 6-wide matrix
    out = torch.full((x.shape[0], 6), 9.0, device="cuda")
    cs.zscore_into(xd, mean, std, out, col0=1)
    z_ref = O.zscore(x, mean.cpu().numpy(), std.cpu().numpy())
    got = out.cpu().numpy()
    assert np.all(got[:, 0] == 9.0) and np.all(got[:, 5] == 9.0)
    assert np.allclose(got[:, 1:5], z_ref, rtol=2e-7, atol=0) and np.all(got[:, 4] == 0.0)
    one = torch.as_tensor([[1.0], [3.0]], device="cuda")
    clamp = cs.zscore_into(one, torch.as_tensor([2.0], dtype=torch.float64, device="cuda"),
                           torch.zeros(1, dtype=torch.float64, device="cuda"), torch.empty((2, 1), device="cuda"))
    assert clamp[0, 0].item() == -np.finfo(np.float32).max and clamp[1, 0].item() == np.finfo(np.float32).max
    # keyed noise: only the positive-mean columns move, by the oracle's draws
    noisy = cs.add_noise_device(xd.clone(), 0.01, seed=1234, stream_id=2, mean=mean)
    ref = O.add_noise_keyed(x, m_ref, 0.01, 1234, 2)
    got = noisy.cpu().numpy()
    assert np.array_equal(got[:, 1], x[:, 1])                           # negative mean: untouched (helper_funcs.py:14)
    for c in (0, 2, 3):
        # device Box-Muller runs on v_log/v_sqrt/v_cos (~1e-6 relative) and the sum rounds to fp32
        tol = 2e-5 * abs(m_ref[c]) * 0.01 + 1.2e-7 * np.abs(x[:, c])
        assert np.all(np.abs(got[:, c] - ref[:, c]) <= tol)
    # every column-count instantiation of the two elementwise kernels (1..4 are compile-time, anything else run-time)
    for cols in (1, 2, 3, 5):
        xc = (rng.normal(size=(7001, cols)) * 2.0 + 3.0).astype(np.float32)
        xcd = torch.as_tensor(xc, device="cuda")
        mc, sc = cs.column_stats(xcd)
        outc = cs.zscore_into(xcd, mc, sc, torch.empty((7001, cols), device="cuda")).cpu().numpy()
        assert np.allclose(outc, O.zscore(xc, mc.cpu().numpy(), sc.cpu().numpy()), rtol=2e-7, atol=0), cols
        nz = cs.add_noise_device(xcd.clone(), 0.02, seed=9, stream_id=1, mean=mc).cpu().numpy()
        refn = O.add_noise_keyed(xc, mc.cpu().numpy(), 0.02, 9, 1)
        assert np.all(np.abs(nz - refn) <= 2e-5 * 3.0 * 0.02 * 4 + 1.2e-7 * np.abs(xc)), cols
    assert abs((got[:, 0] - x[:, 0]).std() / (0.01 * m_ref[0]) - 1.0) < 0.02


@pytest.mark.parametrize("cx,cy,rows", [(2, 1, 100003), (3, 1, 65537), (4, 2, 7001), (5, 3, 7001), (1, 1, 5), (2, 1, 0)])
def test_zscore_concat_equals_two_zscores(ssc, cx, cy, rows):
    """The one-pass network-input matrix (NND_MB_agent.py:303-318) is bit-identical to z-scoring dataX and dataY into it
    with two launches, including zero-std columns (NaN -> 0, +-inf -> +-largest finite) -- and matches the oracle."""
    from smartstartcontinuous_amd import collect_samples as cs
    rng = np.random.default_rng(cx * 10 + cy)
    x = (rng.normal(size=(rows, cx)) * 3.0 + 1.0).astype(np.float32)
    y = (rng.normal(size=(rows, cy)) * 0.5 - 2.0).astype(np.float32)
    if rows > 10:
        y[:, 0] = 4.0                                                    # a constant column: std 0
    xd, yd = torch.as_tensor(x, device="cuda"), torch.as_tensor(y, device="cuda")
    if rows == 0:
        z = torch.zeros(1, dtype=torch.float64, device="cuda")
        assert cs.zscore_concat(xd, z.expand(cx).contiguous(), z.expand(cx).contiguous(), yd, z, z).shape == (0, cx + cy)
        return
    (mx, sx), (my, sy) = cs.column_stats(xd), cs.column_stats(yd)
    two = torch.full((rows, cx + cy), 7.0, device="cuda")
    cs.zscore_into(xd, mx, sx, two, 0)
    cs.zscore_into(yd, my, sy, two, cx)
    one = cs.zscore_concat(xd, mx, sx, yd, my, sy)
    assert torch.equal(one, two)
    ref = np.concatenate([O.zscore(x, mx.cpu().numpy(), sx.cpu().numpy()), O.zscore(y, my.cpu().numpy(), sy.cpu().numpy())], axis=1)
    assert np.allclose(one.cpu().numpy(), ref, rtol=2e-7, atol=0)
    if rows > 10:
        assert np.all(one.cpu().numpy()[:, cx] == 0.0)


def test_collect_samples_end_to_end(ssc):
    """perform_rollouts / CollectSamples with the reference's signature on the stock MountainCar env: list-of-arrays
    result, every transition obeys the env, and the device data set equals the host formatting of those lists."""
    from smartstartcontinuous_amd import collect_samples as cs
    env = ssc.make("MountainCarContinuous-v0", device="cuda", seed=77)
    states, controls, starts, _ = cs.perform_rollouts(cs.Policy_Random(env), 25, 333, False, cs.CollectSamples, env, 3, 1)
    assert len(states) == 25 and all(s.shape == (len(c), 2) and c.shape[1] == 1 for s, c in zip(states, controls))
    assert all(1 <= len(s) <= 333 for s in states) and all(np.array_equal(s[0], st) for s, st in zip(states, starts))
    assert all(-0.6 <= s[0, 0] <= -0.4 and s[0, 1] == 0 for s in states)          # fresh resets (:84-86)
    assert len({float(s[0, 0]) for s in states}) == 25
    for s, c in zip(states, controls):
        assert np.all(np.abs(c) <= 1.0)
        p2, v2, _, d = O.mc_step(s[:-1, 0], s[:-1, 1], c[:-1, 0])
        assert np.max(np.abs(p2 - s[1:, 0])) < 2.4e-7 and np.max(np.abs(v2 - s[1:, 1])) < 1e-8 and not d.any()
    # same seed, same call number => same rollouts; formatted on the device
    ts = cs.CollectSamples(env, cs.Policy_Random(env)).collect_dataset(25, 333)
    X, Y = cs.generate_training_data_inputs(states, controls)
    Z = cs.generate_training_data_outputs([s.astype(np.float32) for s in states])
    assert np.array_equal(ts.dataX.cpu().numpy(), X.astype(np.float32))
    assert np.array_equal(ts.dataY.cpu().numpy(), Y.astype(np.float32))
    assert np.array_equal(ts.dataZ.cpu().numpy(), Z)
    # a second collection of the same collector is a different one (validation set)
    c2 = cs.CollectSamples(env, cs.Policy_Random(env))
    a = c2.collect_samples(4, 50)[0]
    b = c2.collect_samples(4, 50)[0]
    assert not np.array_equal(a[0], b[0])


def test_collect_samples_stops_at_terminal_steps(ssc):
    """With TimeLimit(40) every rollout is cut at 40 steps (terminal = done OR time limit, as TimeLimit.step reports)."""
    from smartstartcontinuous_amd import collect_samples as cs
    env = ssc.Continuous_MountainCarEnv_Editted.make_timed_env(1.0, max_episode_steps=40, seed=5)
    states, controls, _, _ = cs.CollectSamples(env, cs.Policy_Random(env)).collect_samples(70, 100)
    assert all(len(s) == 40 for s in states)
    ts = cs.CollectSamples(env, cs.Policy_Random(env)).collect_dataset(70, 100)
    assert len(ts) == 70 * 39 and ts.lens.eq(40).all()


def test_nnd_mb_agent_collects_and_trains_its_own_data(ssc):
    """NND_MB_agent(env, ...) without training_data runs the reference constructor's collection branch
    (NND_MB_agent.py:215-319) on the device; training on it gives a model that predicts held-out rollouts."""
    from smartstartcontinuous_amd.agents import NND_MB_agent
    env = ssc.make("MountainCarContinuous-v0", device="cuda", seed=3)
    agent = NND_MB_agent(env, None, num_fc_layers=1, depth_fc_layers=32, precision="f32", seed=3,
                         num_rollouts_train=25, steps_per_rollout_train=333, num_rollouts_val=20,
                         steps_per_rollout_val=333)
    rows = agent.dataX.shape[0]
    assert rows == 25 * 332 and agent._train_inputs.shape == (rows, 3) and agent._train_outputs.shape == (rows, 2)
    assert len(agent.states_val) == 20 and agent.states_val[0].shape == (333, 2)
    # the z-scored set is what NND_MB_agent.py:302-319 leaves: zero mean, unit std per column
    zi, zo = agent._train_inputs.double(), agent._train_outputs.double()
    assert zi.mean(0).abs().max().item() < 1e-5 and (zi.std(0, unbiased=False) - 1).abs().max().item() < 1e-4
    assert zo.mean(0).abs().max().item() < 1e-5 and (zo.std(0, unbiased=False) - 1).abs().max().item() < 1e-4
    nm = agent.dyn_model.norm
    assert abs(nm.mean_y[0]) < 0.05 and abs(nm.std_y[0] - 1 / np.sqrt(3)) < 0.02     # U(-1, 1) actions
    # one-step prediction on the validation rollouts the constructor collected, before and after training
    S, A = np.stack(agent.states_val), np.stack(agent.controls_val)            # [20, 333, 2], [20, 333, 1]
    s0 = S[:, :-1].reshape(-1, 2).astype(np.float32)
    act = A[:, :-1].reshape(-1, 1, 1).astype(np.float32)
    true_next = S[:, 1:].reshape(-1, 2)

    def one_step_err():
        pred = agent.dyn_model.do_forward_sim(torch.as_tensor(s0, device="cuda"), torch.as_tensor(act, device="cuda"))
        return np.abs(pred[1].cpu().numpy() - true_next).mean(axis=0) / np.abs(true_next - s0).mean(axis=0)
    before = one_step_err()
    last_loss = agent.train_dynamics_model(nEpoch=12, fraction_use_new=0.0, rng=np.random.RandomState(0))
    after = one_step_err()
    assert last_loss < 0.05 and (after < 0.35).all() and (after < 0.3 * before).all(), (before, after, last_loss)


def test_mpc_rollout_chunks_aggregate_into_a_retraining(ssc):
    """The vectorised counterpart of NND_MB_agent.train_dynamics_model (:437-480): transitions logged while the
    navigator drives the envs become the "new" rows -- formatted and z-scored with the INITIAL statistics on the
    device -- and are mixed into Dyn_Model.train's batches.  Nothing is copied to the host."""
    from smartstartcontinuous_amd import collect_samples as cs, navigator as nav, numerical as num
    from smartstartcontinuous_amd.agents import init_dynamics_weights
    env1 = ssc.make("MountainCarContinuous-v0", device="cuda", seed=21)
    col = cs.CollectSamples(env1, cs.Policy_Random(env1))
    ts = col.collect_dataset(64, 200)
    (mx, sx), (my, sy), (mz, sz) = (cs.column_stats(v) for v in (ts.dataX, ts.dataY, ts.dataZ))
    host = lambda t: t.cpu().numpy()
    norm = dict(mean_x=host(mx), std_x=host(sx), mean_y=host(my), std_y=host(sy), mean_z=host(mz), std_z=host(sz))
    old_in, old_out = ts.normalised(norm)
    assert old_in.shape == (len(ts), 3) and abs(old_in.double().mean(0)).max().item() < 1e-5
    Ws, bs = init_dynamics_weights(3, 2, 1, 32, torch.Generator().manual_seed(3))
    model = nav.DynamicsModel(Ws, bs, norm, 2, 1, precision="f32")
    model.train(old_in, old_out, np.zeros((0, 3)), np.zeros((0, 2)), 6, 0.0, rng=np.random.RandomState(0))
    # P envs follow recorded paths under MPC; their logged transitions are the aggregated data
    states, _, _, _ = col.collect_samples(8, 120)
    wps, lefts, radii = [], [], []
    for path in states:
        stds, means = num.path_deltas_stds_and_means_per_dim(path)
        rad = num.radii_calc(means, stds, 1, 1, 1)
        dist = num.elliptical_euclidean_distance_function_generator(rad)
        wps.append(path); radii.append(rad); lefts.append(num.distances_left(path, dist))
    batch = nav.NavigatorBatch(model, nav.MpcProblemSet(wps, lefts, radii, [0] * 8), num_control_samples=512, horizon=4, seed=5)
    venv = ssc.VecEnv("MountainCarContinuous-v0", 8, seed=23)
    venv.reset()
    chunk = venv.rollout(60, policy=ssc.MpcPolicy(batch))
    new = cs.dataset_from_chunk(chunk)
    assert len(new) == 8 * 59 and torch.equal(new.dataX[:59], chunk.obs[:, :59, 0].t())
    new_in, new_out = new.normalised(norm)
    ref_in = O.zscore(np.concatenate([host(new.dataX), host(new.dataY)], 1),
                      np.concatenate([norm["mean_x"], norm["mean_y"]]), np.concatenate([norm["std_x"], norm["std_y"]]))
    assert np.allclose(host(new_in), ref_in, rtol=2e-7, atol=0)
    w_before = [w.clone() for w in model.W]
    loss = model.train(old_in, old_out, new_in, new_out, 2, 0.5, rng=np.random.RandomState(1))
    assert np.isfinite(loss) and loss < 0.1 and not torch.equal(w_before[0], model.W[0])


def test_dataset_full_size_properties(ssc):
    """BASELINE-size collection (65 536 Pendulum rollouts x 200 steps = 13 M rows): sizes add up, rows inside a
    rollout chain (x + z of one row is the x of the next), spot rows equal the chunk."""
    from smartstartcontinuous_amd import collect_samples as cs
    n, K = 65536, 200
    env = ssc.VecEnv("Pendulum-v0", n, device="cuda", seed=9)
    chunk = env.rollout(K, policy=ssc.RandomPolicy())
    ts = cs.dataset_from_chunk(chunk)
    assert ts.lens.eq(K).all() and len(ts) == n * (K - 1)       # TimeLimit(200): the only terminal step is the last
    X = ts.dataX.view(n, K - 1, 3); Z = ts.dataZ.view(n, K - 1, 3); Y = ts.dataY.view(n, K - 1)
    assert torch.equal(X, chunk.obs[:, :K - 1].permute(2, 1, 0)) and torch.equal(Y, chunk.act[:K - 1].t())
    assert torch.equal(Z, (chunk.obs2[:, :K - 1] - chunk.obs[:, :K - 1]).permute(2, 1, 0))
    nxt = X[:, :-1] + Z[:, :-1]
    assert (nxt - X[:, 1:]).abs().max().item() <= 1e-6
    mean, std = cs.column_stats(ts.dataY)
    assert abs(mean.item()) < 2e-3 and abs(std.item() - 4 / np.sqrt(12)) < 2e-3       # U(-2, 2)
