"""GPU parity tests of the SmartStart navigator kernels through the C ABI: dynamics-MLP
forward, forward simulation, MPC action sampling, trajectory scoring (incl. the batch-global
projection quirk) and action selection -- BASELINE config 4 at test sizes."""
import numpy as np
import pytest

from oracle import ssc_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nav():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: the HIP path has no fallback")
    from smartstartcontinuous_amd import navigator
    from smartstartcontinuous_amd import _ffi
    _ffi.lib()
    return navigator


def make_mlp(rng, dims):
    """xavier-normal weights AND biases (feedforward_network.py:8,14-23)."""
    Ws = [rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1])) for i in range(len(dims) - 1)]
    bs = [rng.normal(size=dims[i + 1]) * np.sqrt(2.0 / (1 + dims[i + 1])) for i in range(len(dims) - 1)]
    return [w.astype(np.float32) for w in Ws], [b.astype(np.float32) for b in bs]


def make_norm(rng, d, a):
    return dict(mean_x=rng.normal(size=d) * 0.3, std_x=rng.uniform(0.05, 1.0, d), mean_y=rng.normal(size=a) * 0.1,
                std_y=rng.uniform(0.3, 1.2, a), mean_z=rng.normal(size=d) * 0.01, std_z=rng.uniform(0.005, 0.05, d))


@pytest.mark.parametrize("dims", [(3, 32, 2), (4, 500, 500, 3), (3, 500, 2), (12, 64, 48, 40, 8)])
def test_mlp_forward_f32(nav, dims):
    rng = np.random.default_rng(sum(dims))
    Ws, bs = make_mlp(rng, dims)
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, dims[-1], dims[0] - dims[-1] if dims[0] > dims[-1] else 1),
                              state_dim=dims[-1], act_dim=max(1, dims[0] - dims[-1]))
    for m in (1, 15, 16, 17, 1000, 4099):
        x = rng.normal(size=(m, dims[0])).astype(np.float32)
        ref = O.mlp_forward(x, Ws, bs)
        got = model.forward(x, precision="f32").cpu().numpy()
        scale = np.maximum(1.0, np.abs(ref).max())
        assert np.max(np.abs(got - ref)) <= 1e-4 * scale, (dims, m)       # 1e-4 rel (fp32)


@pytest.mark.parametrize("dims,n,batchsize", [((3, 32, 2), 1300, 512), ((4, 500, 500, 3), 2048, 512), ((3, 500, 2), 700, 100),
                                               ((12, 64, 48, 40, 8), 515, 512)])
def test_run_validation_vs_oracle(nav, dims, n, batchsize):
    """Dyn_Model.run_validation (dynamics_model.py:173-196): the mean of the per-batch MSE over the full batches only, in
    fp32 and through the MFMA forward where the network has one; fewer rows than one batch is the reference's division by zero."""
    rng = np.random.default_rng(n + sum(dims))
    Ws, bs = make_mlp(rng, dims)
    d = dims[-1]
    a = max(1, dims[0] - d)
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, d, a), state_dim=d, act_dim=a)
    X = rng.normal(size=(n, dims[0])).astype(np.float32)
    Z = (O.mlp_forward(X, Ws, bs) + rng.normal(size=(n, d)) * 0.3).astype(np.float32)
    Z[n // batchsize * batchsize:] += 100.0                 # the ragged tail must not count
    ref = O.dyn_run_validation(X.astype(np.float64), Z.astype(np.float64), Ws, bs, batchsize)
    got = model.run_validation(X, Z, batchsize=batchsize)
    assert isinstance(got, float) and abs(got - ref) <= 1e-5 * ref, (got, ref)
    assert abs(model.run_validation(torch.as_tensor(X, device="cuda"), torch.as_tensor(Z, device="cuda"), batchsize=batchsize) - got) == 0.0
    if len(dims) <= 4 and dims[0] <= 10:
        assert abs(model.run_validation(X, Z, batchsize=batchsize, precision="bf16_mfma") - ref) <= 2e-2 * ref
    with pytest.raises(ZeroDivisionError):
        model.run_validation(X[:batchsize - 1], Z[:batchsize - 1], batchsize=batchsize)


@pytest.mark.parametrize("dims,H", [((3, 32, 2), 4), ((4, 500, 500, 3), 4), ((3, 500, 2), 20)])
def test_forward_sim_f32(nav, dims, H):
    rng = np.random.default_rng(7 + H)
    d, a = dims[-1], dims[0] - dims[-1]
    Ws, bs = make_mlp(rng, dims)
    norm = make_norm(rng, d, a)
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=a)
    m = 777
    A = rng.uniform(-1, 1, size=(m, H, a)).astype(np.float32)
    s0 = rng.normal(size=d).astype(np.float32) * 0.3
    S = model.do_forward_sim(s0, A).cpu().numpy()
    ref = O.dyn_forward_sim(s0, A, norm32(norm), Ws, bs)
    assert S.shape == (H + 1, m, d) and np.array_equal(S[0], np.tile(s0, (m, 1)))
    assert np.max(np.abs(S - ref)) <= 1e-4 * np.maximum(1.0, np.abs(ref).max())
    # per-row start states
    s0m = rng.normal(size=(m, d)).astype(np.float32) * 0.3
    S = model.do_forward_sim(s0m, A).cpu().numpy()
    ref = O.dyn_forward_sim(s0m, A, norm32(norm), Ws, bs)
    assert np.max(np.abs(S - ref)) <= 1e-4 * np.maximum(1.0, np.abs(ref).max())


def norm32(norm):
    """the kernel receives fp32 statistics; give the oracle the same values"""
    return {k: np.asarray(v, np.float32).astype(np.float64) for k, v in norm.items()}


def test_forward_sim_zero_std_quirk(nav):
    """std == 0 (e.g. the velocity column of a data set that never moved): nan_to_num gives 0 for 0/0
    and +-max for x/0 (dynamics_model.py:228-229)."""
    rng = np.random.default_rng(3)
    Ws, bs = make_mlp(rng, (3, 32, 2))
    Ws[0][1, :] = 0.0     # the second input is multiplied by 0 so the +-max never reaches the output as inf
    norm = make_norm(rng, 2, 1)
    norm["std_x"] = np.array([0.3, 0.0])
    norm["mean_x"] = np.array([0.0, 0.0])
    norm["mean_z"] = np.array([0.001, 0.0])
    norm["std_z"] = np.array([0.01, 0.0])
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=2, act_dim=1)
    A = rng.uniform(-1, 1, size=(64, 3, 1)).astype(np.float32)
    S = model.do_forward_sim(np.array([0.1, 0.0], np.float32), A).cpu().numpy()
    ref = O.dyn_forward_sim(np.array([0.1, 0.0], np.float32), A, norm32(norm), Ws, bs)
    assert np.isfinite(S).all() and np.max(np.abs(S - ref)) <= 1e-5


def test_mpc_sample_actions_bit_exact(nav):
    for (P, N, H, low, high) in [(3, 50, 4, [-1.0], [1.0]), (2, 17, 5, [-2.0, 0.0], [2.0, 1.0]), (1, 5000, 20, [-1.0], [1.0])]:
        A = nav.mpc_sample_actions(P, N, H, low, high, seed=99, problem_id0=7, t=11).cpu().numpy()
        assert A.shape == (P * N, H, len(low))
        for p in range(P):
            ref = O.mpc_action_samples(99, 7 + p, N, H, len(low), 11, low, high)
            assert np.array_equal(A[p * N:(p + 1) * N], ref)


def _score_case(nav, S, wps, lefts, radii, cur, per_row=False, **kw):
    """S [H+1, P, N, d] fp64 -> kernel scores / best vs oracle per problem"""
    H1, P, N, d = S.shape
    ps = nav.MpcProblemSet(wps, lefts, radii, cur, per_row_projection=per_row, **kw)
    S32 = torch.as_tensor(S.reshape(H1, P * N, d), dtype=torch.float32, device="cuda")
    scores, best, best_score = nav.mpc_score(ps, S32)
    scores, best, best_score = scores.cpu().numpy(), best.cpu().numpy(), best_score.cpu().numpy()
    for p in range(P):
        Sp = S32[:, p * N:(p + 1) * N].cpu().numpy().astype(np.float64)
        ref, ref_best_score, ref_best, _ = O.mpc_scores_add_delta(
            Sp, np.asarray(wps[p], np.float32), np.asarray(lefts[p], np.float32), np.asarray(radii[p], np.float32),
            cur[p], per_row_projection=per_row, **{k: v for k, v in kw.items() if k in ("theta", "gamma")},
            hpf=kw.get("horizontal_penalty_factor", 0.5))
        tol = 1e-3 * np.maximum(1.0, np.abs(ref).max())          # scores <= 1e-3 rel
        assert np.max(np.abs(scores[p] - ref)) <= tol, p
        # the chosen sample's reference score is within tolerance of the reference max (not index equality)
        assert ref[best[p]] >= ref_best_score - tol
        assert abs(best_score[p] - scores[p, best[p]]) == 0 and scores[p, best[p]] == scores[p].max()
        assert best[p] == int(np.argmax(scores[p]))              # lowest index on ties
    return scores


def test_mpc_score_reference_kats(nav, golden_dir):
    """Scores produced by the reference's own numerical.py helpers (tests/golden/make_goldens.py)."""
    kats = np.load(f"{golden_dir}/numerical_kats.npz")
    for c in range(int(kats["n_mpc"])):
        S, wp, left, radii = (kats[f"m{c}_{k}"] for k in ("S", "wp", "left", "radii"))
        H1, N, d = S.shape
        scores = _score_case(nav, S.reshape(H1, 1, N, d), [wp], [left], [radii], [int(kats[f"m{c}_cur"])])
        ref = kats[f"m{c}_scores"]
        assert np.max(np.abs(scores[0] - ref)) <= 1e-3 * np.maximum(1.0, np.abs(ref).max())
        assert ref[int(np.argmax(scores[0]))] >= ref.max() - 1e-3 * np.maximum(1.0, np.abs(ref).max())


@pytest.mark.parametrize("per_row,P,N,H,Wmax", [(False, 5, 1000, 4, 60), (True, 5, 1000, 4, 60), (False, 3, 37, 4, 60),
                                                (False, 2, 8192, 4, 60), (False, 2, 9001, 4, 60), (True, 2, 9001, 4, 60),
                                                (False, 70, 16, 4, 60), (True, 33, 9, 4, 60), (False, 21, 32, 7, 60),
                                                (False, 19, 33, 9, 60), (False, 9, 64, 4, 60), (False, 7, 65, 4, 60),
                                                (False, 3, 300, 4, 6000), (False, 5, 48, 12, 6000)])
def test_mpc_score_multi_problem(nav, per_row, P, N, H, Wmax):
    """N <= 64: the one-launch kernel (a problem inside one wave, 16 / 32 / 64 lanes); larger N the two-pass path.
    Only a window of H + 4 waypoints around the current one is staged, so plans of any length work (Wmax = 6000:
    more than the 48 KB LDS carve of the first versions held)."""
    rng = np.random.default_rng(11)
    d = 2
    wps, lefts, radii, cur = [], [], [], []
    S = np.empty((H + 1, P, N, d))
    for p in range(P):
        W = int(rng.integers(2, Wmax))
        wp = np.cumsum(rng.normal(scale=[0.02, 0.004], size=(W, d)), axis=0) + [-0.5, 0.0]
        stds, means = O.path_deltas_stds_and_means_per_dim(wp) if W > 2 else (np.array([0.01, 0.002]), np.array([0.02, 0.004]))
        r = O.radii_calc(means, stds, 1, 1, 1) + 1e-4
        wps.append(wp); radii.append(r); lefts.append(O.distances_left(wp, O.distance_func(r)))
        c = int(rng.integers(0, W)); cur.append(c)
        s = wp[c] + rng.normal(scale=r * 0.7)
        S[0, p] = s
        for t in range(H):
            S[t + 1, p] = S[t, p] + rng.normal(scale=r * 0.9, size=(N, d))
    _score_case(nav, S, wps, lefts, radii, cur, per_row=per_row, theta=1.0, gamma=0.75, horizontal_penalty_factor=0.5)


def test_mpc_pipeline_select_action(nav):
    """sample -> forward sim -> score -> select: the navigator's get_action for P envs at once."""
    rng = np.random.default_rng(21)
    P, N, H, d, a = 4, 500, 4, 2, 1
    Ws, bs = make_mlp(rng, (3, 32, 2))
    norm = dict(mean_x=[-0.5, 0.0], std_x=[0.2, 0.02], mean_y=[0.0], std_y=[0.6], mean_z=[0.0, 0.0], std_z=[0.01, 0.002])
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=a)
    A = nav.mpc_sample_actions(P, N, H, [-1.0], [1.0], seed=5, problem_id0=100, t=3)
    states = rng.normal(size=(P, d)).astype(np.float32) * [0.1, 0.01] + [-0.5, 0.0]
    s0 = torch.as_tensor(np.repeat(states, N, axis=0), dtype=torch.float32, device="cuda")
    S = model.do_forward_sim(s0, A)
    wps = [np.cumsum(rng.normal(scale=[0.02, 0.004], size=(12, d)), axis=0) + states[p] for p in range(P)]
    radii = [np.array([0.03, 0.006])] * P
    lefts = [O.distances_left(w, O.distance_func(radii[0])) for w in wps]
    ps = nav.MpcProblemSet(wps, lefts, radii, [0] * P)
    scores, best, _ = nav.mpc_score(ps, S)
    action, path = nav.mpc_select_action(A, S, best, P, noise_amount=0.005, seed=5, problem_id0=100, t=3)
    action, path, best = action.cpu().numpy(), path.cpu().numpy(), best.cpu().numpy()
    An, Sn = A.cpu().numpy(), S.cpu().numpy()
    for p in range(P):
        row = p * N + best[p]
        g = O.mpc_noise_gaussian(5, np.array([100 + p], np.uint64), 3, 0)[0]
        assert abs(action[p, 0] - (An[row, 0, 0] + 0.005 * g)) <= 1e-6
        assert np.array_equal(path[p], Sn[:, row])
    clean, _ = nav.mpc_select_action(A, S, torch.as_tensor(best, device="cuda"), P, 0.0, 5, 100, 3, want_path=False)
    assert np.array_equal(clean.cpu().numpy()[:, 0], An[np.arange(P) * N + best, 0, 0])


# ------------------------------------------------------------------ bf16 MFMA path --
MFMA_SHAPES = [(3, 32, 2), (4, 500, 500, 3), (3, 500, 2), (4, 100, 100, 3), (12, 64, 8), (5, 20, 20, 2)]


@pytest.mark.parametrize("dims", MFMA_SHAPES)
def test_mlp_forward_mfma(nav, dims):
    rng = np.random.default_rng(sum(dims) + 1)
    Ws, bs = make_mlp(rng, dims)
    d = dims[-1]
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, d, max(1, dims[0] - d)), state_dim=d, act_dim=max(1, dims[0] - d))
    for m in (1, 31, 64, 255, 257, 3000):
        x = rng.normal(size=(m, dims[0])).astype(np.float32)
        got = model.forward(x, precision="bf16_mfma").cpu().numpy()
        ref = O.mlp_forward(x, Ws, bs)
        emu = O.mlp_forward_bf16emu(x, Ws, bs)
        scale = np.maximum(1.0, np.abs(ref).max())
        assert np.max(np.abs(got - ref)) <= 3e-2 * scale, (dims, m)      # 3e-2 rel (bf16)
        assert np.max(np.abs(got - emu)) <= 2e-3 * scale, (dims, m)      # same roundings, different sum order
        if m > 1:   # a mean over out_dim values is just the max again
            assert np.mean(np.abs(got - emu)) <= 1e-4 * scale, (dims, m)


def test_mfma_weight_layout_one_hot(nav):
    """Exact check of the fragment packing: one-hot weights route a single input through chosen hidden
    units; any row/column or k-permutation mix-up shows up as an exact mismatch."""
    rng = np.random.default_rng(0)
    for trial in range(6):
        depth, in_dim, out_dim = 500, 4, 3
        Ws = [np.zeros((in_dim, depth), np.float32), np.zeros((depth, depth), np.float32), np.zeros((depth, out_dim), np.float32)]
        bs = [np.zeros(depth, np.float32), np.zeros(depth, np.float32), np.zeros(out_dim, np.float32)]
        i, u, v, o = int(rng.integers(in_dim)), int(rng.integers(depth)), int(rng.integers(depth)), int(rng.integers(out_dim))
        Ws[0][i, u] = 1.0
        Ws[1][u, v] = 0.5
        Ws[2][v, o] = 2.0
        bs[2][:] = [0.25, -0.5, 1.0]
        model = nav.DynamicsModel(Ws, bs, make_norm(rng, 3, 1), state_dim=3, act_dim=1)
        # bf16-representable inputs: layer 1 (bf16 head + residual products) is then exact
        x = O.round_bf16(rng.uniform(0.1, 2.0, size=(300, in_dim))).astype(np.float32)
        got = model.forward(x, precision="bf16_mfma").cpu().numpy()
        ref = np.tile(bs[2], (300, 1)).astype(np.float64)
        ref[:, o] += 2.0 * O.round_bf16(0.5 * O.round_bf16(x[:, i]).astype(np.float64)).astype(np.float64)
        assert np.max(np.abs(got - ref)) < 1e-6, (trial, i, u, v, o)


@pytest.mark.parametrize("dims,H", [((3, 32, 2), 4), ((4, 500, 500, 3), 4), ((3, 500, 2), 20), ((4, 128, 128, 3), 7)])
def test_forward_sim_mfma(nav, dims, H):
    rng = np.random.default_rng(17 + H)
    d, a = dims[-1], dims[0] - dims[-1]
    Ws, bs = make_mlp(rng, dims)
    norm = make_norm(rng, d, a)
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=a, precision="bf16_mfma")
    for m in (1, 300, 1000):
        A = rng.uniform(-1, 1, size=(m, H, a)).astype(np.float32)
        for s0 in (rng.normal(size=d).astype(np.float32) * 0.3, rng.normal(size=(m, d)).astype(np.float32) * 0.3):
            S = model.do_forward_sim(s0, A).cpu().numpy()
            ref = O.dyn_forward_sim(s0, A, norm32(norm), Ws, bs)
            emu = O.dyn_forward_sim(s0, A, norm32(norm), Ws, bs, forward=O.mlp_forward_bf16emu)
            scale = np.maximum(1.0, np.abs(ref).max())
            assert S.shape == (H + 1, m, d)
            assert np.array_equal(S[0], np.broadcast_to(s0, (m, d)))
            assert np.max(np.abs(S - ref)) <= 3e-2 * scale, (dims, H, m)
            assert np.max(np.abs(S - emu)) <= 3e-3 * scale, (dims, H, m)
    # fp32 and MFMA paths agree on the same inputs
    Sf = model.do_forward_sim(s0, A, precision="f32").cpu().numpy()
    assert np.max(np.abs(Sf - S)) <= 3e-2 * np.maximum(1.0, np.abs(Sf).max())


@pytest.mark.parametrize("dims,B", [((3, 32, 2), 512), ((4, 500, 500, 3), 512), ((3, 40, 24, 16, 2), 77),
                                    ((4, 500, 3), 512), ((4, 20, 3), 77), ((9, 100, 5), 100), ((3, 512, 2), 33),
                                    ((12, 7, 8), 1)])
def test_mlp_train_step_kernel_vs_oracle(nav, dims, B):
    """ssc_mlp_train_step (forward, MSE, backprop, tf-style Adam) for several steps against the fp64 oracle.
    One hidden layer runs the fused one-launch kernel (all three instantiations, unit counts on and off the
    power-of-two padding, batches that do not fill the last 32-row block); the others the generic chain."""
    rng = np.random.default_rng(sum(dims))
    Ws, bs = make_mlp(rng, dims)
    d, a = dims[-1], dims[0] - dims[-1]
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, d, max(a, 1)), state_dim=d, act_dim=max(a, 1))
    n = 2000
    X = rng.normal(size=(n, dims[0])).astype(np.float32)
    Z = (rng.normal(size=(n, dims[-1])) * 0.5).astype(np.float32)
    Xd, Zd = torch.as_tensor(X, device="cuda"), torch.as_tensor(Z, device="cuda")
    oW, ob = [w.astype(np.float64) for w in Ws], [b.astype(np.float64) for b in bs]
    adam = dict(mW=[np.zeros_like(w) for w in oW], vW=[np.zeros_like(w) for w in oW],
                mb=[np.zeros_like(b) for b in ob], vb=[np.zeros_like(b) for b in ob], t=0)
    loss = torch.zeros(1, device="cuda")
    for step in range(4):
        idx = rng.permutation(n)[:B].astype(np.int32)
        oW, ob, adam, ref_loss = O.mlp_train_step(oW, ob, adam, X[idx], Z[idx], lr=1e-3)
        model.train_step(Xd, Zd, torch.as_tensor(idx, device="cuda"), lr=1e-3, loss=loss)
        assert abs(loss.item() - ref_loss) <= 2e-4 * max(1.0, ref_loss), (step, loss.item(), ref_loss)
    for l in range(len(oW)):
        # 4 Adam steps of ~1e-3: parameters agree to a small fraction of one step
        assert np.max(np.abs(model.W[l].cpu().numpy() - oW[l])) <= 2e-5, l
        assert np.max(np.abs(model.b[l].cpu().numpy() - ob[l])) <= 2e-5, l
    assert int(model._adam["t"].item()) == 4


@pytest.mark.parametrize("dims", [(3, 32, 2), (4, 500, 3), (4, 24, 24, 3)])
def test_mlp_train_steps_equals_single_steps(nav, dims):
    """ssc_mlp_train_steps: n consecutive steps enqueued by one call == the same steps one call at a time, bit for
    bit on both paths (every reduction has a fixed order; there is no float atomic anywhere), losses included."""
    rng = np.random.default_rng(7)
    n, B, steps = 3000, 512, 6
    X = torch.as_tensor(rng.normal(size=(n, dims[0])).astype(np.float32), device="cuda")
    Z = torch.as_tensor((rng.normal(size=(n, dims[-1])) * 0.5).astype(np.float32), device="cuda")
    idx = torch.as_tensor(np.stack([rng.permutation(n)[:B] for _ in range(steps)]).astype(np.int32), device="cuda")
    d, a = dims[-1], dims[0] - dims[-1]
    Ws, bs = make_mlp(rng, dims)
    m1 = nav.DynamicsModel(Ws, bs, make_norm(rng, d, a), state_dim=d, act_dim=a)
    m2 = nav.DynamicsModel(Ws, bs, make_norm(rng, d, a), state_dim=d, act_dim=a)
    losses = m1.train_steps(X, Z, idx, lr=1e-3)
    one = torch.zeros(1, device="cuda")
    for k in range(steps):
        m2.train_step(X, Z, idx[k], lr=1e-3, loss=one)
        assert one.item() == losses[k].item()
    for l in range(len(Ws)):
        assert torch.equal(m1.W[l], m2.W[l]) and torch.equal(m1.b[l], m2.b[l])
    assert losses[-1].item() < losses[0].item()


def test_dynamics_model_training_learns_mountaincar(nav, golden_dir):
    """Semantic check: a 1x32 model trained on the reference's dataX/Y/Z reaches a low one-step
    error on the reference's held-out validation rollouts (states_val / controls_val)."""
    import smartstartcontinuous_amd as ssc
    from smartstartcontinuous_amd.agents import NND_MB_agent
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    env = ssc.make("MountainCarContinuous-v0")
    agent = NND_MB_agent(env, None, horizon=4, num_control_samples=100, num_fc_layers=1, depth_fc_layers=32,
                         training_data=dict(dataX=g["dataX"], dataY=g["dataY"], dataZ=g["dataZ"]), precision="f32", seed=3)
    S, A = g["states_val"], g["controls_val"]
    s0 = S[:, :-1].reshape(-1, 2).astype(np.float32)
    act = A[:, :-1].reshape(-1, 1, 1).astype(np.float32)
    true_next = S[:, 1:].reshape(-1, 2)

    def one_step_err():
        pred = agent.dyn_model.do_forward_sim(torch.as_tensor(s0, device="cuda"), torch.as_tensor(act, device="cuda"))
        return np.abs(pred[1].cpu().numpy() - true_next).mean(axis=0) / np.abs(true_next - s0).mean(axis=0)
    before = one_step_err()
    np.random.seed(0)
    last_loss = agent.train_dynamics_model(nEpoch=12, fraction_use_new=0.0, batchsize=512, lr=0.001)
    after = one_step_err()
    assert last_loss < 0.05 and (after < 0.35).all() and (after < 0.3 * before).all(), (before, after, last_loss)
    # aggregation round as in the reference (train_dynamics_model :437-480): the replay buffer's transitions,
    # add_noise on states and deltas, z-scored with the initial statistics, mixed 50/50 into every batch
    for ep in range(3):
        agent.replay_buffer.start_new_episode(agent)
        for t in range(S.shape[1] - 1):
            agent.replay_buffer.add(agent, S[ep, t], A[ep, t], 0.0, False, S[ep, t + 1])
    xn, zn = agent.aggregated_dataset(np.random.RandomState(1))
    assert xn.shape == (3 * (S.shape[1] - 1), 3) and zn.shape == (xn.shape[0], 2)
    clean = (S[:3, :-1].reshape(-1, 2) - np.asarray([agent.dyn_model.norm.mean_x[i] for i in range(2)])) / \
        np.asarray([agent.dyn_model.norm.std_x[i] for i in range(2)])
    # helper_funcs.add_noise only touches columns with a positive mean: position (mean < 0) stays exact
    assert np.allclose(xn[:, 0], clean[:, 0], atol=1e-6)
    loss2 = agent.train_dynamics_model(nEpoch=3, fraction_use_new=0.5, rng=np.random.RandomState(2))
    assert np.isfinite(loss2) and (one_step_err() < 0.35).all()


def test_vecenv_rollout_with_mpc_policy(nav, golden_dir):
    """rollout(K, policy='mpc').  P MountainCar envs each follow their own recorded path with
    the navigator; every logged step is re-derived by the oracle from the state the env was in
    (samples bit-exact, forward sim + scores in fp64, env step, waypoint bookkeeping of
    NND_MB_agent.observe)."""
    import smartstartcontinuous_amd as ssc
    g = np.load(f"{golden_dir}/mc_reference_rollouts.npz")
    rng = np.random.default_rng(5)
    P, N, H, K, seed = 3, 400, 4, 7, 31
    Ws, bs = make_mlp(rng, (3, 32, 2))
    from smartstartcontinuous_amd.agents import NND_MB_agent
    nm = NND_MB_agent.normalisation_from_data(g["dataX"], g["dataY"], g["dataZ"])
    model = nav.DynamicsModel(Ws, bs, nm, state_dim=2, act_dim=1, precision="f32")
    env = ssc.VecEnv("MountainCarContinuous-v0", P, seed=seed)
    env.reset()
    wps, lefts, radii = [], [], []
    for p in range(P):
        path = g["states_val"][p, 10:60]
        stds, means = O.path_deltas_stds_and_means_per_dim(path)
        r = O.radii_calc(means, stds, 1, 1, 1)
        w = O.waypoints_from_path(O.path_shortcutter(path, O.distance_func(r), 1))
        wps.append(w); radii.append(r); lefts.append(O.distances_left(w, O.distance_func(r)))
    env.s0.copy_(torch.as_tensor([w[0][0] for w in wps], dtype=torch.float32))
    env.s1.copy_(torch.as_tensor([w[0][1] for w in wps], dtype=torch.float32))
    ps = nav.MpcProblemSet(wps, lefts, radii, [0] * P)
    batch = nav.NavigatorBatch(model, ps, num_control_samples=N, horizon=H, seed=seed, steps_before_giving_up_on_waypoint=2)
    chunk = env.rollout(K, ssc.MpcPolicy(batch))
    torch.cuda.synchronize()
    obs = chunk.obs.cpu().numpy(); act = chunk.act.cpu().numpy(); obs2 = chunk.obs2.cpu().numpy()
    nm32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in nm.items()}
    idx = [0] * P
    done_act = [0] * P
    for k in range(K):
        for p in range(P):
            s = obs[:, k, p]
            done_act[p] += 1
            A = O.mpc_action_samples(seed, p, N, H, 1, k, [-1.0], [1.0])
            S = O.dyn_forward_sim(s, A, nm32, Ws, bs)
            scores, best_score, _, _ = O.mpc_scores_add_delta(S, np.asarray(wps[p], np.float32), np.asarray(lefts[p], np.float32),
                                                          np.asarray(radii[p], np.float32), idx[p])
            noise = 0.005 * O.mpc_noise_gaussian(seed, np.array([p], np.uint64), k, 0)[0]
            cand = np.argmin(np.abs(A[:, 0, 0] + noise - act[k, p]))
            assert abs(A[cand, 0, 0] + noise - act[k, p]) <= 1e-6
            assert scores[cand] >= best_score - 1e-3 * max(1.0, abs(best_score))
            p2, v2, _, _ = O.mc_step(s[0], s[1], act[k, p])
            assert abs(p2 - obs2[0, k, p]) <= 2.4e-7 and abs(v2 - obs2[1, k, p]) <= 1e-8
            # NND_MB_agent.observe (:360-373)
            dist = O.distance_func(radii[p])
            W = len(wps[p])
            dc = dist(obs2[:, k, p], wps[p][idx[p]]); dn = dist(obs2[:, k, p], wps[p][min(idx[p] + 1, W - 1)])
            if ((dc <= 1 or dn <= dc) and idx[p] != W - 1) or (done_act[p] > 2 and idx[p] != W - 1):
                idx[p] += 1
                done_act[p] = 0
    assert ps.cur_idx.cpu().tolist() == idx and batch.actions_done.cpu().tolist() == done_act
    assert env.stats.cpu().numpy()[2] == P * K and env.t == K


def test_mfma_prepared_image_matches_per_call_pack_and_tracks_weight_changes(nav):
    """ssc_dyn_prepare + SSC_PREC_BF16_MFMA_PREPARED (what DynamicsModel uses) == the per-call pack of
    SSC_PREC_BF16_MFMA, bit for bit; the image follows set_weights / set_norm / train_step."""
    import ctypes
    from smartstartcontinuous_amd import _ffi
    rng = np.random.default_rng(5)
    dims, H, m = (4, 500, 500, 3), 3, 700
    Ws, bs = make_mlp(rng, dims)
    norm = make_norm(rng, 3, 1)
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=3, act_dim=1, precision="bf16_mfma")
    A = torch.as_tensor(rng.uniform(-1, 1, size=(m, H, 1)).astype(np.float32), device="cuda")
    s0 = torch.as_tensor(rng.normal(size=(m, 3)).astype(np.float32) * 0.3, device="cuda")

    def per_call_pack():
        lib = _ffi.lib()
        S = torch.empty((H + 1, m, 3), dtype=torch.float32, device="cuda")
        nbytes = lib.ssc_dyn_workspace_bytes(ctypes.byref(model.desc), m, _ffi.SSC_PREC_BF16_MFMA)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        _ffi.check(lib.ssc_dyn_forward_sim(ctypes.byref(model.desc), ctypes.byref(model.norm), m, H, 3, 1,
                                           _ffi.ptr(s0), m, _ffi.ptr(A), _ffi.ptr(S), _ffi.SSC_PREC_BF16_MFMA,
                                           _ffi.ptr(ws), ws.numel(), torch.cuda.current_stream().cuda_stream))
        return S.cpu().numpy()

    S1 = model.do_forward_sim(s0, A).cpu().numpy()
    assert not model._image_stale and np.array_equal(S1, per_call_pack())
    assert np.array_equal(model.do_forward_sim(s0, A).cpu().numpy(), S1)          # image reused
    # new weights -> new image
    Ws2, bs2 = make_mlp(rng, dims)
    model.set_weights(Ws2, bs2)
    S2 = model.do_forward_sim(s0, A).cpu().numpy()
    assert not np.array_equal(S2, S1) and np.array_equal(S2, per_call_pack())
    ref = O.dyn_forward_sim(s0.cpu().numpy(), A.cpu().numpy(), norm32(norm), Ws2, bs2)
    assert np.max(np.abs(S2 - ref)) <= 3e-2 * max(1.0, np.abs(ref).max())
    # new statistics -> new image
    norm2 = make_norm(rng, 3, 1)
    model.set_norm(norm2)
    S3 = model.do_forward_sim(s0, A).cpu().numpy()
    assert not np.array_equal(S3, S2) and np.array_equal(S3, per_call_pack())
    # a training step moves the weights in place -> the image follows
    X = torch.as_tensor(rng.normal(size=(64, 4)).astype(np.float32), device="cuda")
    Z = torch.as_tensor(rng.normal(size=(64, 3)).astype(np.float32), device="cuda")
    model.train_step(X, Z, torch.arange(64, dtype=torch.int32, device="cuda"), lr=0.01)
    S4 = model.do_forward_sim(s0, A).cpu().numpy()
    assert not np.array_equal(S4, S3) and np.array_equal(S4, per_call_pack())


def test_default_mfma_precision_falls_back_to_fp32_kernels_for_uncovered_networks(nav):
    """A model whose DEFAULT precision is bf16_mfma but whose shape the MFMA kernel does not cover (3 hidden
    layers) runs on the fp32 GPU kernels; asking for MFMA explicitly still raises."""
    from smartstartcontinuous_amd import _ffi
    rng = np.random.default_rng(2)
    dims = (3, 40, 40, 40, 2)
    Ws, bs = make_mlp(rng, dims)
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, 2, 1), state_dim=2, act_dim=1, precision="bf16_mfma")
    x = rng.normal(size=(100, 3)).astype(np.float32)
    got = model.forward(x).cpu().numpy()
    assert np.max(np.abs(got - O.mlp_forward(x, Ws, bs))) <= 1e-4
    with pytest.raises(_ffi.SscError):
        model.forward(x, precision="bf16_mfma")
    # 12 inputs with two 500-unit layers: also outside the MFMA kernel (one layer-1 k-step holds 10)
    dims = (12, 500, 500, 8)
    Ws, bs = make_mlp(rng, dims)
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, 8, 4), state_dim=8, act_dim=4, precision="bf16_mfma")
    x = rng.normal(size=(50, 12)).astype(np.float32)
    ref = O.mlp_forward(x, Ws, bs)
    assert np.max(np.abs(model.forward(x).cpu().numpy() - ref)) <= 1e-4 * max(1.0, np.abs(ref).max())
    # 10 inputs: covered (KIN = 10 instantiation)
    dims = (10, 500, 500, 7)
    Ws, bs = make_mlp(rng, dims)
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, 7, 3), state_dim=7, act_dim=3, precision="bf16_mfma")
    x = rng.normal(size=(300, 10)).astype(np.float32)
    ref = O.mlp_forward(x, Ws, bs)
    got = model.forward(x).cpu().numpy()
    assert model._mfma_ok and np.max(np.abs(got - ref)) <= 3e-2 * max(1.0, np.abs(ref).max())
    emu = O.mlp_forward_bf16emu(x, Ws, bs)
    assert np.max(np.abs(got - emu)) <= 2e-3 * max(1.0, np.abs(ref).max())
    # ... and as a forward simulation with 7 state + 3 action dims
    A = rng.uniform(-1, 1, size=(200, 3, 3)).astype(np.float32)
    s0 = rng.normal(size=(200, 7)).astype(np.float32) * 0.3
    nm = make_norm(rng, 7, 3)
    model = nav.DynamicsModel(Ws, bs, nm, state_dim=7, act_dim=3, precision="bf16_mfma")
    S = model.do_forward_sim(s0, A).cpu().numpy()
    ref = O.dyn_forward_sim(s0, A, norm32(nm), Ws, bs)
    assert np.max(np.abs(S - ref)) <= 3e-2 * max(1.0, np.abs(ref).max())


def test_forward_sim_and_scoring_full_size_properties(nav):
    """BASELINE config 4 at full size (65 536 rows, in 4 / 2x500 / out 3, H = 4; 16 problems x 4096 samples):
    S[0] is the start state, a row's trajectory depends on its own (s0, actions) only (a row permutation permutes
    the output bit for bit), 512 rows spread over the batch match the fp64 oracle, the scorer's argmax is the max
    of its scores and identical problems get identical results."""
    rng = np.random.default_rng(44)
    P, N, H, d, a = 16, 4096, 4, 3, 1
    M = P * N
    Ws, bs = make_mlp(rng, (4, 500, 500, 3))
    norm = make_norm(rng, d, a)
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=a, precision="bf16_mfma")
    A = torch.as_tensor(rng.uniform(-2, 2, size=(M, H, a)).astype(np.float32), device="cuda")
    s0 = torch.as_tensor((rng.normal(size=(M, d)) * 0.3).astype(np.float32), device="cuda")
    S = model.do_forward_sim(s0, A)
    assert S.shape == (H + 1, M, d) and torch.equal(S[0], s0) and bool(torch.isfinite(S).all())
    perm = torch.as_tensor(rng.permutation(M), device="cuda")
    Sp = model.do_forward_sim(s0[perm], A[perm])
    assert torch.equal(Sp, S[:, perm])
    rows = np.sort(rng.choice(M, 512, replace=False))
    ref = O.dyn_forward_sim(s0[rows].cpu().numpy(), A[rows].cpu().numpy(), norm32(norm), Ws, bs)
    got = S[:, rows].cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 3e-2 * max(1.0, np.abs(ref).max())       # 3e-2 rel (bf16)
    # scoring: every problem follows the same 60-waypoint path; problems 0 and 1 get identical samples
    wp = np.cumsum(rng.normal(scale=[0.02, 0.004, 0.01], size=(60, d)), axis=0)
    stds, means = O.path_deltas_stds_and_means_per_dim(wp)
    r = O.radii_calc(means, stds, 1, 1, 1) + 1e-4
    left = O.distances_left(wp, O.distance_func(r))
    ps = nav.MpcProblemSet([wp] * P, [left] * P, [r] * P, [0] * P, theta=1.0, gamma=0.75, horizontal_penalty_factor=0.5)
    S4 = (torch.as_tensor(wp[0], dtype=torch.float32, device="cuda") + 0.02 * S).contiguous()
    S4[:, N:2 * N] = S4[:, :N]
    scores, best, best_score = nav.mpc_score(ps, S4)
    scores = scores.view(P, N)
    assert bool(torch.isfinite(scores).all())
    assert torch.equal(best_score, scores.max(dim=1).values) and torch.equal(best.long(), scores.argmax(dim=1))
    assert torch.equal(scores[0], scores[1]) and int(best[0]) == int(best[1])
    p0 = S4[:, :N].cpu().numpy().astype(np.float64)
    ref_scores, _, _, _ = O.mpc_scores_add_delta(p0, wp.astype(np.float32), np.asarray(left, np.float32),
                                                 np.asarray(r, np.float32), 0, theta=1.0, gamma=0.75, hpf=0.5)
    assert np.max(np.abs(scores[0].cpu().numpy() - ref_scores)) <= 1e-3 * max(1.0, np.abs(ref_scores).max())


@pytest.mark.parametrize("env_name,max_steps,prec", [("MountainCarContinuous-v0", 9, "f32"), ("Pendulum-v0", 6, "bf16_mfma")])
def test_mpc_rollout_graph_replay_equals_step_by_step_path(nav, env_name, max_steps, prec):
    """rollout(K, MpcPolicy): the HIP-graph path (fused ssc_mpc_rollout_step, device step counters) produces the
    same bits as the step-by-step path through the single-purpose entry points -- log, env state, navigator
    state, statistics, episode records -- across auto-resets (short time limit) and over two consecutive chunks."""
    import smartstartcontinuous_amd as ssc
    rng = np.random.default_rng(8)
    P, N, H, K = 70, 128, 3, 8
    d = 2 if env_name.startswith("Mountain") else 3
    Ws, bs = make_mlp(rng, (d + 1, 32, d))
    norm = make_norm(rng, d, 1)
    paths = [np.cumsum(rng.normal(scale=0.02, size=(12, d)), axis=0) for _ in range(P)]

    def setup(graph):
        env = ssc.VecEnv(env_name, P, seed=21, max_episode_steps=max_steps)
        env.reset()
        model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=1, precision=prec)
        wps, lefts, radii = [], [], []
        for pth in paths:
            stds, means = O.path_deltas_stds_and_means_per_dim(pth)
            r = O.radii_calc(means, stds, 1, 1, 1) + 1e-3
            wps.append(pth); radii.append(r); lefts.append(O.distances_left(pth, O.distance_func(r)))
        ps = nav.MpcProblemSet(wps, lefts, radii, [1] * P)
        lo, hi = ([-1.0], [1.0]) if d == 2 else ([-2.0], [2.0])
        batch = nav.NavigatorBatch(model, ps, num_control_samples=N, horizon=H, action_low=lo, action_high=hi, seed=4,
                                   steps_before_giving_up_on_waypoint=2)
        return env, ps, batch, ssc.MpcPolicy(batch, graph=graph)

    out = []
    for graph in (True, False):
        env, ps, batch, pol = setup(graph)
        ring = ssc.EpisodeRing(4096, "cuda") if graph else None
        chunks = []
        for c in range(2):
            ch = env.rollout(K, pol, ring=ring) if graph else env.rollout(K, pol)
            torch.cuda.synchronize()
            chunks.append({k: getattr(ch, k).clone() for k in ("obs", "act", "rew", "done", "obs2")})
        out.append((env, ps, batch, chunks, ring))
    (eg, pg, bg, cg, ring), (ee, pe, be, ce, _) = out
    for c in range(2):
        for key in ("obs", "act", "rew", "done", "obs2"):
            assert torch.equal(cg[c][key], ce[c][key]), (c, key)
    assert torch.equal(eg.s0, ee.s0) and torch.equal(eg.s1, ee.s1) and torch.equal(eg.steps, ee.steps)
    assert torch.equal(eg.ep_ret, ee.ep_ret) and eg.t == ee.t == 2 * K
    assert torch.equal(pg.cur_idx, pe.cur_idx) and torch.equal(bg.actions_done, be.actions_done)
    sg, se = eg.stats.cpu().numpy(), ee.stats.cpu().numpy()
    assert sg[2] == se[2] == 2 * K * P and sg[3] == se[3] and sg[1] == se[1] and abs(sg[0] - se[0]) <= 1e-6 * max(1.0, abs(se[0]))
    n_done = int(sum(int(ch["done"].sum()) for ch in cg))
    assert n_done >= P and sg[3] == n_done                                     # the short time limit forces resets
    (ids, lens, rets), dropped = ring.drain()
    assert dropped == 0 and len(lens) == n_done and lens.max() <= max_steps


def test_mpc_rollout_graph_follows_weight_updates(nav):
    """The replayed graph reads the CURRENT dynamics model: after train_step (in place) and after set_weights (new
    tensors) the graph path still equals the step-by-step path."""
    import smartstartcontinuous_amd as ssc
    rng = np.random.default_rng(12)
    P, N, H, K, d = 40, 128, 3, 5, 2
    Ws, bs = make_mlp(rng, (3, 500, 500, 2))
    norm = make_norm(rng, d, 1)
    paths = [np.cumsum(rng.normal(scale=[0.02, 0.004], size=(12, d)), axis=0) + [-0.5, 0.0] for _ in range(P)]
    X = torch.as_tensor(rng.normal(size=(64, 3)).astype(np.float32), device="cuda")
    Z = torch.as_tensor(rng.normal(size=(64, 2)).astype(np.float32), device="cuda")
    Ws2, bs2 = make_mlp(rng, (3, 500, 500, 2))

    def run(graph):
        env = ssc.VecEnv("MountainCarContinuous-v0", P, seed=2)
        env.reset()
        model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=1, precision="bf16_mfma")
        wps, lefts, radii = [], [], []
        for pth in paths:
            stds, means = O.path_deltas_stds_and_means_per_dim(pth)
            r = O.radii_calc(means, stds, 1, 1, 1) + 1e-3
            wps.append(pth); radii.append(r); lefts.append(O.distances_left(pth, O.distance_func(r)))
        batch = nav.NavigatorBatch(model, nav.MpcProblemSet(wps, lefts, radii, [0] * P), num_control_samples=N, horizon=H, seed=6)
        pol = ssc.MpcPolicy(batch, graph=graph)
        acts = [env.rollout(K, pol).act.clone()]
        for _ in range(3):
            model.train_step(X, Z, torch.arange(64, dtype=torch.int32, device="cuda"), lr=0.05)   # in place
        acts.append(env.rollout(K, pol).act.clone())
        model.set_weights(Ws2, bs2)                                                                # new tensors
        acts.append(env.rollout(K, pol).act.clone())
        torch.cuda.synchronize()
        return acts
    g, e = run(True), run(False)
    for i in range(3):
        assert torch.equal(g[i], e[i]), i
    assert not torch.equal(g[0], g[1]) and not torch.equal(g[1], g[2])


def test_mpc_rollout_graphs_live_on_the_navigator(nav):
    """a HIP graph bakes the navigator's N / H and buffer addresses in.  A NEW NavigatorBatch over the same
    problems, model, env and chunk -- other num_control_samples, other horizon, possibly at a recycled id() -- must not
    replay its predecessor's graph: every variant equals its own step-by-step path."""
    import gc
    import smartstartcontinuous_amd as ssc
    rng = np.random.default_rng(3)
    P, K, d = 24, 4, 2
    Ws, bs = make_mlp(rng, (3, 32, 2))
    norm = make_norm(rng, d, 1)
    paths = [np.cumsum(rng.normal(scale=[0.02, 0.004], size=(10, d)), axis=0) + [-0.5, 0.0] for _ in range(P)]
    wps, lefts, radii = [], [], []
    for pth in paths:
        stds, means = O.path_deltas_stds_and_means_per_dim(pth)
        r = O.radii_calc(means, stds, 1, 1, 1) + 1e-3
        wps.append(pth); radii.append(r); lefts.append(O.distances_left(pth, O.distance_func(r)))
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=1, precision="f32")

    def run(graph, shapes, env, ps, chunk):
        acts = []
        for (N, H) in shapes:
            batch = nav.NavigatorBatch(model, ps, num_control_samples=N, horizon=H, seed=5)
            env.rollout(K, ssc.MpcPolicy(batch, graph=graph), out=chunk)
            acts.append(chunk.act.clone())
            del batch
            gc.collect()                      # the next NavigatorBatch may land on the same id()
        torch.cuda.synchronize()
        return acts
    shapes = [(64, 3), (192, 3), (64, 5), (320, 2), (64, 3)]
    out = []
    for graph in (True, False):
        env = ssc.VecEnv("MountainCarContinuous-v0", P, seed=8)
        env.reset()
        ps = nav.MpcProblemSet(wps, lefts, radii, [0] * P)
        out.append(run(graph, shapes, env, ps, ssc.TransitionChunk(2, K, P, "cuda")))
    for i in range(len(shapes)):
        assert torch.equal(out[0][i], out[1][i]), shapes[i]
    assert not torch.equal(out[0][0], out[0][1])


@pytest.mark.parametrize("dims,prec,P,N,H,act", [((4, 500, 500, 3), "bf16_mfma", 3, 1000, 4, 1), ((3, 32, 2), "bf16_mfma", 2, 333, 5, 1),
                                                 ((5, 64, 64, 3), "bf16_mfma", 2, 700, 7, 2), ((3, 32, 2), "f32", 2, 100, 6, 1),
                                                 ((12, 64, 8), "bf16_mfma", 1, 515, 3, 4)])
def test_forward_sim_with_in_kernel_sampling_equals_sample_then_sim(nav, dims, prec, P, N, H, act):
    """ssc_mpc_forward_sim (the forward simulation draws its candidate action sequences itself) == ssc_mpc_sample_actions
    followed by ssc_dyn_forward_sim, bit for bit: the trajectories, and the action matrix when it is asked for -- for
    horizons x action widths that need 1, 2 and 3 Philox calls per row, ragged row counts, and through a device step
    counter (d_t_base)."""
    rng = np.random.default_rng(5)
    d = dims[-1]
    assert dims[0] == d + act
    Ws, bs = make_mlp(rng, dims)
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, d, act), state_dim=d, act_dim=act, precision=prec)
    low, high = [-1.0, -0.5, 0.0, -2.0][:act], [1.0, 0.5, 3.0, 2.0][:act]
    M = P * N
    s0 = torch.as_tensor((rng.normal(size=(P, d)) * 0.2).astype(np.float32), device="cuda")
    t_base = torch.full((1,), 40, dtype=torch.int64, device="cuda")
    for (t, tb) in ((7, None), (3, t_base)):
        A = nav.mpc_sample_actions(P, N, H, low, high, seed=77, problem_id0=9, t=t, t_base=tb)
        S_ref = model.do_forward_sim(s0, A).clone()
        sp = nav.mpc_sampling(N, low, high, 77, 9, t, t_base=tb)
        A_out = torch.full((M, H, act), -7.0, device="cuda")
        S = model.do_forward_sim_sampled(s0, sp, M, H, A_out=A_out)
        assert torch.equal(S, S_ref) and torch.equal(A_out, A)
        if prec != "f32":
            S2 = model.do_forward_sim_sampled(s0, sp, M, H)          # no action matrix at all
            assert torch.equal(S2, S_ref)


def test_score_select_equals_score_then_select(nav):
    """ssc_mpc_score_select (pass B's last block also selects) == ssc_mpc_score + ssc_mpc_select_action: same scores, winner,
    action (with the 0.005 N(0,1) noise) and predicted path -- with the action matrix given, and with the winner's first
    action regenerated from the sampling specification."""
    rng = np.random.default_rng(31)
    P, N, H, d = 5, 600, 4, 3
    Ws, bs = make_mlp(rng, (d + 1, 32, d))
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, d, 1), state_dim=d, act_dim=1)
    states = (rng.normal(size=(P, d)) * 0.1).astype(np.float32)
    wps = [np.cumsum(rng.normal(scale=0.02, size=(15, d)), axis=0) + states[p] for p in range(P)]
    radii = [np.array([0.03, 0.03, 0.03])] * P
    lefts = [O.distances_left(w, O.distance_func(radii[0])) for w in wps]
    ps = nav.MpcProblemSet(wps, lefts, radii, [0] * P)
    t_base = torch.full((1,), 5, dtype=torch.int64, device="cuda")
    for (t, tb) in ((11, None), (2, t_base)):
        t_eff = t + (0 if tb is None else 5)
        A = nav.mpc_sample_actions(P, N, H, [-2.0], [2.0], seed=8, problem_id0=50, t=t, t_base=tb)
        S = model.do_forward_sim(torch.as_tensor(states, device="cuda"), A)
        scores, best, _ = nav.mpc_score(ps, S)
        action, path = nav.mpc_select_action(A, S, best, P, 0.005, 8, 50, t_eff)
        sp = nav.mpc_sampling(N, [-2.0], [2.0], 8, 50, t, t_base=tb)
        if tb is None:
            sc1, b1, a1, p1 = nav.mpc_score_select(ps, S, A=A, noise_amount=0.005, seed=8, problem_id0=50, t=t_eff)
            assert torch.equal(sc1, scores) and torch.equal(b1, best) and torch.equal(a1, action) and torch.equal(p1, path)
        sc2, b2, a2, p2 = nav.mpc_score_select(ps, S, sampling=sp, noise_amount=0.005, seed=8, problem_id0=50, t=t)
        assert torch.equal(sc2, scores) and torch.equal(b2, best) and torch.equal(a2, action) and torch.equal(p2, path)
    with pytest.raises(nav._ffi.SscError):
        nav.mpc_score_select(ps, S, noise_amount=0.0)                # neither A nor a sampling spec


def test_mpc_rollout_baseline_config4_one_navigator_per_env(nav):
    """BASELINE configs[3] AS WRITTEN: Pendulum-v1, NND_MB dynamics MLP 2x500 on the bf16 MFMA, 65 536 envs, every
    env navigating its own plan (P = 65 536 MPC problems x 16 candidate sequences = 1 Mi simulated rows, H = 4)
    through rollout(K, 'mpc').  The graph path must equal the step-by-step path bit for bit, and a slice of problems
    (the first, the last, and a spread in between) is re-derived by the oracle: candidate samples bit-exact, the
    forward simulation at the bf16 tolerance, the scores of the kernel's own trajectories at 1e-3 (incl. the
    batch-global projection sums over exactly the 16 samples of a problem), the executed action, the Pendulum-v1
    step and the waypoint bookkeeping."""
    import smartstartcontinuous_amd as ssc
    rng = np.random.default_rng(65536)
    P, N, H, K, d, seed, pid0 = 65536, 16, 4, 3, 3, 77, 5
    Ws, bs = make_mlp(rng, (4, 500, 500, 3))
    Ws[-1] *= 0.05
    bs[-1] *= 0.05                                              # small deltas: simulated states stay near the plan
    norm = dict(mean_x=[0.0, 0.0, 0.0], std_x=[0.7, 0.7, 3.0], mean_y=[0.0], std_y=[1.2], mean_z=[0.0, 0.0, 0.0], std_z=[0.05, 0.05, 0.4])
    # 64 recorded Pendulum paths (oracle rollouts under random torques), tiled over the envs
    base, W = [], 30
    for b in range(64):
        th, thd = rng.uniform(-np.pi, np.pi), rng.uniform(-1, 1)
        pts = []
        for _ in range(W):
            pts.append([np.cos(th), np.sin(th), thd])
            th, thd, _, _ = O.pend_step(th, thd, rng.uniform(-2, 2), v1_order=True)
        base.append(np.asarray(pts))
    radii_b, left_b = [], []
    for pth in base:
        stds, means = O.path_deltas_stds_and_means_per_dim(pth)
        r = O.radii_calc(means, stds, 1, 1, 1) + 1e-3
        radii_b.append(r); left_b.append(O.distances_left(pth, O.distance_func(r)))
    which = np.arange(P) % 64
    wp = np.stack(base)[which].reshape(P * W, d).astype(np.float32)
    left = np.stack(left_b)[which].reshape(-1).astype(np.float32)
    radii = np.stack(radii_b)[which].astype(np.float32)
    off = (np.arange(P + 1) * W).astype(np.int32)
    cur0 = (np.arange(P) % 5).astype(np.int32)                  # plans in progress: windows start at different waypoints

    def setup(graph):
        env = ssc.VecEnv("Pendulum-v1", P, seed=seed)
        assert env.params.pend_v1_order == 1
        env.reset()
        start = np.stack(base)[which, cur0]                     # every env sits at its current waypoint
        env.s0.copy_(torch.as_tensor(np.arctan2(start[:, 1], start[:, 0]), dtype=torch.float32))
        env.s1.copy_(torch.as_tensor(start[:, 2], dtype=torch.float32))
        model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=1, precision="bf16_mfma")
        ps = nav.MpcProblemSet.from_packed(wp, left, off, radii, cur0)
        batch = nav.NavigatorBatch(model, ps, num_control_samples=N, horizon=H, action_low=[-2.0], action_high=[2.0],
                                   seed=seed, problem_id0=pid0, steps_before_giving_up_on_waypoint=2)
        return env, ps, batch, ssc.MpcPolicy(batch, graph=graph)

    eg, pg, bg, polg = setup(True)
    c1 = eg.rollout(K - 1, polg)
    cur_before = pg.cur_idx.cpu().numpy().copy()
    done_before = bg.actions_done.cpu().numpy().copy()
    c2 = eg.rollout(1, polg)
    torch.cuda.synchronize()
    ee, pe, be, pole = setup(False)
    d1 = ee.rollout(K - 1, pole)
    d2 = ee.rollout(1, pole)
    torch.cuda.synchronize()
    for a, b in ((c1, d1), (c2, d2)):
        for key in ("obs", "act", "rew", "done", "obs2"):
            assert torch.equal(getattr(a, key), getattr(b, key)), key
    assert torch.equal(eg.s0, ee.s0) and torch.equal(eg.s1, ee.s1) and torch.equal(pg.cur_idx, pe.cur_idx)
    assert torch.equal(bg.actions_done, be.actions_done)
    assert eg.stats.cpu().numpy()[2] == P * K and eg.t == K
    assert int(pg.cur_idx.max()) > int(cur0.max())              # plans advanced
    # ---- oracle slice of the LAST step (t = K - 1) ----
    fb = bg._fused_buffers(eg.device)
    Sk = bg._S.cpu().numpy()                                    # [H+1, P*N, d] trajectories of the last step
    scores_k = fb["scores"].cpu().numpy().reshape(P, N)
    best_k = fb["best"].cpu().numpy()
    obs, act, obs2 = (x.cpu().numpy() for x in (c2.obs, c2.act, c2.obs2))
    nm32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in norm.items()}
    t = K - 1
    for p in [0, 1, 63, 64, 4097, 32768, 50001, 65534, 65535] + rng.integers(0, P, 23).tolist():
        s = obs[:, 0, p].astype(np.float64)
        A = O.mpc_action_samples(seed, pid0 + p, N, H, 1, t, [-2.0], [2.0])
        rows = slice(p * N, (p + 1) * N)
        assert np.array_equal(Sk[0, rows], np.broadcast_to(obs[:, 0, p], (N, d)))
        ref = O.dyn_forward_sim(s, A, nm32, Ws, bs)
        assert np.max(np.abs(Sk[:, rows] - ref)) <= 3e-2 * max(1.0, np.abs(ref).max()), p
        b = which[p]
        sc, best_score, _, _ = O.mpc_scores_add_delta(Sk[:, rows].astype(np.float64), base[b].astype(np.float32),
                                                      left_b[b].astype(np.float32), radii_b[b].astype(np.float32), int(cur_before[p]))
        tol = 1e-3 * max(1.0, np.abs(sc).max())
        assert np.max(np.abs(scores_k[p] - sc)) <= tol, p
        assert best_k[p] == int(np.argmax(scores_k[p])) and sc[best_k[p]] >= best_score - tol
        noise = 0.005 * O.mpc_noise_gaussian(seed, np.array([pid0 + p], np.uint64), t, 0)[0]
        assert abs(A[best_k[p], 0, 0] + noise - act[0, p]) <= 1e-6
        th = np.arctan2(s[1], s[0])
        th2, thd2, _, _ = O.pend_step(th, s[2], act[0, p], v1_order=True)
        assert np.max(np.abs(obs2[:, 0, p] - [np.cos(th2), np.sin(th2), thd2])) <= 3e-6
        # NND_MB_agent.observe (:360-373)
        dist = O.distance_func(radii_b[b])
        i0, da = int(cur_before[p]), int(done_before[p]) + 1
        dc = dist(obs2[:, 0, p], base[b][i0]); dn = dist(obs2[:, 0, p], base[b][min(i0 + 1, W - 1)])
        if ((dc <= 1 or dn <= dc) and i0 != W - 1) or (da > 2 and i0 != W - 1):
            i0, da = i0 + 1, 0
        assert int(pg.cur_idx[p]) == i0 and int(bg.actions_done[p]) == da, p


@pytest.mark.parametrize("prec,dims", [("f32", (3, 32, 2)), ("f32", (4, 100, 3)), ("bf16_mfma", (3, 32, 2)), ("bf16_mfma", (4, 64, 64, 3))])
def test_forward_sim_skips_masked_problems(nav, prec, dims):
    """ssc_mpc_sampling.d_problem_active: the rows of live problems come out exactly as without a mask (and as the oracle
    computes them), the rows of masked problems are left alone wherever the kernel can skip them -- the fused fp32 kernel
    per row, the resident-weight MFMA kernels per wave (rows of a masked problem that share a wave with a live one are
    still computed, which is allowed)."""
    rng = np.random.default_rng(12)
    d, act = dims[-1], dims[0] - dims[-1]
    P, N, H = 37, 16, 3
    Ws, bs = make_mlp(rng, dims)
    norm = make_norm(rng, d, act)
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=act, precision=prec)
    s0 = torch.as_tensor((rng.normal(size=(P, d)) * 0.2).astype(np.float32), device="cuda")
    low, high = [-1.0] * act, [1.0] * act
    M = P * N
    full = model.do_forward_sim_sampled(s0, nav.mpc_sampling(N, low, high, 5, 3, 11), M, H,
                                       A_out=torch.zeros((M, H, act), device="cuda")).clone()
    active = torch.as_tensor((rng.random(P) < 0.4).astype(np.uint8), device="cuda")
    active[0] = 1
    out = torch.full((H + 1, M, d), 123.0, device="cuda")
    A_out = torch.zeros((M, H, act), device="cuda")
    got = model.do_forward_sim_sampled(s0, nav.mpc_sampling(N, low, high, 5, 3, 11, active=active), M, H, out=out, A_out=A_out)
    rows_live = active.bool().repeat_interleave(N)
    assert torch.equal(got[:, rows_live], full[:, rows_live])
    dead = got[:, ~rows_live]
    untouched = (dead == 123.0).all(dim=0).all(dim=-1)
    if prec == "f32":
        assert bool(untouched.all())                      # per-row skip
    else:
        assert bool(untouched.any())                      # whole waves of masked problems were skipped
        computed = ~untouched
        assert torch.equal(dead[:, computed], full[:, ~rows_live][:, computed])
    # and the live rows are what the oracle computes
    nm64 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in norm.items()}
    p = int(torch.nonzero(active)[1]) if int(active.sum()) > 1 else 0
    A = O.mpc_action_samples(5, 3 + p, N, H, act, 11, low, high)
    ref = O.dyn_forward_sim(s0[p].cpu().numpy(), A, nm64, Ws, bs)                   # [H+1, N, d]
    tol = 1e-4 if prec == "f32" else 3e-2
    g = got[:, p * N:(p + 1) * N].cpu().numpy()
    assert np.max(np.abs(g - ref)) <= tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dims,P,N,H", [((3, 32, 2), 5, 7, 4), ((3, 30, 2), 5, 7, 9), ((4, 32, 3), 3, 401, 5), ((4, 50, 3), 6, 33, 4), ((3, 128, 2), 1, 1001, 2)])
def test_fused_fp32_simulation_two_rows_per_lane(nav, dims, P, N, H):
    """dyn_small_sim_pair_kernel (the navigators' shapes: state 2 or 3, one action): two rows per lane -- odd candidate
    counts (a pair straddles two problems), odd row totals (the last lane has one row), depths that are and are not a
    multiple of 4 (scalar-load and LDS-image weight paths), 2 and 3 Philox calls per row, masked problems, a compact work
    list, actions from memory, one start state per problem / per row / for all rows -- against the fp64 oracle at 1e-5 and,
    between the variants, bit for bit."""
    rng = np.random.default_rng(sum(dims) + N)
    d = dims[-1]
    Ws, bs = make_mlp(rng, dims)
    norm = make_norm(rng, d, 1)
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=1, precision="f32")
    nm64 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in norm.items()}
    M = P * N
    s0 = torch.as_tensor((rng.normal(size=(P, d)) * 0.2).astype(np.float32), device="cuda")
    sp = nav.mpc_sampling(N, [-1.0], [1.0], 19, 4, 6)
    A_out = torch.full((M, H, 1), -7.0, device="cuda")
    S = model.do_forward_sim_sampled(s0, sp, M, H, A_out=A_out).clone()
    for p in range(P):
        A = O.mpc_action_samples(19, 4 + p, N, H, 1, 6, [-1.0], [1.0])
        assert np.array_equal(A_out[p * N:(p + 1) * N].cpu().numpy(), A.astype(np.float32))
        ref = O.dyn_forward_sim(s0[p].cpu().numpy(), A, nm64, Ws, bs)
        assert np.max(np.abs(S[:, p * N:(p + 1) * N].cpu().numpy() - ref)) <= 1e-5 * max(1.0, np.abs(ref).max())
    # the same candidates read from memory: same bits
    assert torch.equal(model.do_forward_sim(s0, A_out), S)
    # one start state per ROW (rows_per_state = 1) and one for all rows
    s_rows = s0.repeat_interleave(N, dim=0).contiguous()
    assert torch.equal(model.do_forward_sim(s_rows, A_out), S)
    one = model.do_forward_sim(s0[:1].contiguous(), A_out)
    assert torch.equal(one[:, :N], S[:, :N])
    # masked problems: their rows are left alone, the others do not change
    active = torch.as_tensor((np.arange(P) % 2 == 0).astype(np.uint8), device="cuda")
    out = torch.full((H + 1, M, d), 123.0, device="cuda")
    A2 = torch.full((M, H, 1), -7.0, device="cuda")
    got = model.do_forward_sim_sampled(s0, nav.mpc_sampling(N, [-1.0], [1.0], 19, 4, 6, active=active), M, H, out=out, A_out=A2)
    live = active.bool().repeat_interleave(N)
    assert torch.equal(got[:, live], S[:, live]) and bool((got[:, ~live] == 123.0).all())
    assert torch.equal(A2[live], A_out[live]) and bool((A2[~live] == -7.0).all())
    # a compact work list of the live problems: the same rows again
    lst = torch.nonzero(active).flatten().to(torch.int32)
    n_live = torch.tensor([lst.numel()], dtype=torch.int32, device="cuda")
    live_list = torch.zeros(P, dtype=torch.int32, device="cuda")
    live_list[:lst.numel()] = lst
    out2 = torch.full((H + 1, M, d), 123.0, device="cuda")
    got2 = model.do_forward_sim_sampled(s0, nav.mpc_sampling(N, [-1.0], [1.0], 19, 4, 6, active=active, live_list=live_list, n_live=n_live),
                                        M, H, out=out2)
    assert torch.equal(got2, got)


@pytest.mark.parametrize("P,N,H,sampled", [(5000, 16, 4, True), (5000, 16, 1, True), (1, 70000 + 37, 3, True), (300, 257, 2, False)])
def test_streamed_sim_kernel_walking_over_row_tiles_equals_one_tile_per_block(nav, P, N, H, sampled):
    """More row tiles than CUs: a block of the streamed-W2 kernel (2 x 500) WALKS over row tiles (the W2 ring, its barriers and
    the wave groups' lag run on across tiles; the next tile's start states arrive by LDS-DMA under the current tile's last
    step).  Rows are independent, so the walk must reproduce -- bit for bit -- the same kernel run on sub-batches of at most
    256 tiles, which get a block per tile: per-env start states, ragged last tile, H = 1 (every step is a tile boundary), one
    problem spanning every tile, and the actions-from-memory mode.  (BASELINE configs[3] as written: 65 536 envs x 16
    candidates = 4096 row tiles.)"""
    rng = np.random.default_rng(11)
    d, act = 3, 1
    Ws, bs = make_mlp(rng, (d + act, 500, 500, d))
    model = nav.DynamicsModel(Ws, bs, make_norm(rng, d, act), state_dim=d, act_dim=act, precision="bf16_mfma")
    M = P * N
    assert M > 256 * 256                                               # more tiles than the chip has CUs
    s0 = torch.as_tensor((rng.normal(size=(P, d)) * 0.2).astype(np.float32), device="cuda")
    if sampled:
        sp = nav.mpc_sampling(N, [-2.0], [2.0], 77, 9, 5)
        A_all = torch.empty((M, H, act), device="cuda")
        S = model.do_forward_sim_sampled(s0, sp, M, H, A_out=A_all).clone()
        assert torch.equal(A_all, nav.mpc_sample_actions(P, N, H, [-2.0], [2.0], seed=77, problem_id0=9, t=5))
    else:
        A_all = torch.as_tensor(rng.uniform(-2, 2, size=(M, H, act)).astype(np.float32), device="cuda")
        S = model.do_forward_sim(s0, A_all).clone()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(S).all())
    # the same rows in pieces of whole problems, each <= 65 536 rows (<= 256 tiles: one tile per block, no walk), from memory
    per = max(1, 65536 // N) if P > 1 else 1
    if P == 1:                                                         # one problem: pieces of rows, same start state
        for r0 in range(0, M, 65536):
            r1 = min(M, r0 + 65536)
            ref = model.do_forward_sim(s0, A_all[r0:r1].contiguous())
            assert torch.equal(S[:, r0:r1], ref), (r0, r1)
    else:
        for p0 in range(0, P, per):
            p1 = min(P, p0 + per)
            ref = model.do_forward_sim(s0[p0:p1].contiguous(), A_all[p0 * N:p1 * N].contiguous())
            assert torch.equal(S[:, p0 * N:p1 * N], ref), (p0, p1)
    # and against the fp32 kernels on a strided subset of rows (the same tolerance as test_forward_sim_mfma)
    rows = torch.arange(0, M, 997, device="cuda")
    s0_rows = s0[(rows // N).clamp(max=P - 1)]
    ref32 = model.do_forward_sim(s0_rows, A_all[rows].contiguous(), precision="f32")
    err = (S[:, rows] - ref32).abs().max().item() / max(1.0, ref32.abs().max().item())
    assert err <= 3e-2, err
