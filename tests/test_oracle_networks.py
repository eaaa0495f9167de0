"""Parity-unpinned parts of the oracle (actor, critic, dynamics MLP, forward sim, OU, Pendulum):
the reference needs TensorFlow 1.5 / gym and ships no fixture for them, so the restatement is
cross-checked structurally against independent torch-CPU fp64 implementations."""
import ctypes

import numpy as np
import torch

from oracle import ssc_oracle as O
from tests.gpu_util import actor_weights


def _lin(W, b):
    lin = torch.nn.Linear(W.shape[0], W.shape[1]).double()
    with torch.no_grad():
        lin.weight.copy_(torch.as_tensor(W, dtype=torch.float64).t())   # TF [in,out] -> torch [out,in]
        lin.bias.copy_(torch.as_tensor(b, dtype=torch.float64))
    return lin


def test_actor_matches_torch_stack():
    rng = np.random.default_rng(0)
    for (h1, h2, llt) in [(64, 32, True), (128, 64, False), (200, 100, True)]:
        w = actor_weights(2, h1, h2, seed=h1, w3_scale=0.5)
        obs = rng.uniform(-1, 1, size=(257, 2))
        net = [_lin(w["W1"], w["b1"]), _lin(w["W2"], w["b2"]), _lin(w["W3"], w["b3"])]
        x = torch.relu(net[0](torch.as_tensor(obs)))
        x = net[1](x)
        x = torch.tanh(x) if llt else torch.relu(x)
        ref = torch.tanh(net[2](x)).detach().numpy()
        got = O.actor_forward(obs, **w, last_layer_tanh=llt)
        assert np.allclose(got, ref, rtol=0, atol=1e-14)
        # bf16 emulation stays within the bf16 budget of the exact forward
        emu = O.actor_forward_bf16emu(obs, **w, last_layer_tanh=llt)
        assert np.max(np.abs(emu - got)) < 2e-2


def test_critic_matches_torch_stack():
    rng = np.random.default_rng(1)
    W1, b1 = rng.normal(size=(2, 64)), rng.normal(size=64)
    W2, b2 = rng.normal(size=(65, 32)) * 0.2, rng.normal(size=32)
    W3, b3 = rng.normal(size=(32, 1)) * 0.1, rng.normal(size=1)
    obs, act = rng.normal(size=(50, 2)), rng.uniform(-1, 1, size=(50, 1))
    x = torch.relu(_lin(W1, b1)(torch.as_tensor(obs)))
    x = torch.tanh(_lin(W2, b2)(torch.cat([x, torch.as_tensor(act)], dim=-1)))
    ref = _lin(W3, b3)(x).detach().numpy()
    assert np.allclose(O.critic_forward(obs, act, W1, b1, W2, b2, W3, b3), ref, atol=1e-13)


def _mlp(rng, dims):
    Ws = [rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1])) for i in range(len(dims) - 1)]
    bs = [rng.normal(size=dims[i + 1]) * 0.1 for i in range(len(dims) - 1)]
    return Ws, bs


def test_mlp_and_forward_sim(oracle_clib):
    rng = np.random.default_rng(2)
    for dims in [(3, 32, 2), (4, 500, 500, 3), (3, 500, 2)]:
        Ws, bs = _mlp(rng, dims)
        x = rng.normal(size=(33, dims[0]))
        h = torch.as_tensor(x)
        for W, b in zip(Ws[:-1], bs[:-1]):
            h = torch.relu(_lin(W, b)(h))
        ref = _lin(Ws[-1], bs[-1])(h).detach().numpy()
        got = O.mlp_forward(x, Ws, bs)
        assert np.allclose(got, ref, atol=1e-12)
        # the C restatement
        w = np.concatenate([W.reshape(-1) for W in Ws])
        b = np.concatenate(bs)
        y = np.empty((33, dims[-1]))
        scratch = np.empty(2 * 33 * max(dims))
        d32 = np.asarray(dims, np.int32)
        dp = ctypes.POINTER(ctypes.c_double)
        oracle_clib.ssc_oracle_mlp_forward(ctypes.c_int64(33), ctypes.c_int(len(dims) - 1),
                                           d32.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), w.ctypes.data_as(dp),
                                           b.ctypes.data_as(dp), np.ascontiguousarray(x).ctypes.data_as(dp),
                                           y.ctypes.data_as(dp), scratch.ctypes.data_as(dp))
        assert np.allclose(y, ref, atol=1e-12)
    # forward sim == explicit loop; std==0 normalisation quirk (nan_to_num)
    Ws, bs = _mlp(rng, (3, 32, 2))
    norm = dict(mean_x=np.array([-0.5, 0.0]), std_x=np.array([0.2, 0.0]), mean_y=np.array([0.1]),
                std_y=np.array([0.6]), mean_z=np.array([0.0, 0.0]), std_z=np.array([0.01, 0.001]))
    A = rng.uniform(-1, 1, size=(17, 5, 1))
    S = O.dyn_forward_sim(np.array([-0.5, 0.0]), A, norm, Ws, bs)
    assert S.shape == (6, 17, 2) and np.all(S[0] == [-0.5, 0.0])
    cur = np.tile([-0.5, 0.0], (17, 1))
    for t in range(5):
        xs = np.zeros_like(cur)
        xs[:, 0] = (cur[:, 0] + 0.5) / 0.2
        d = cur[:, 1] - 0.0
        xs[:, 1] = np.where(d == 0, 0.0, np.sign(d) * np.finfo(np.float64).max)   # x/0 -> +-DBL_MAX, 0/0 -> 0
        ys = (A[:, t] - 0.1) / 0.6
        z = O.mlp_forward(np.concatenate([xs, ys], 1), Ws, bs)
        cur = cur + z * norm["std_z"] + norm["mean_z"]
        assert np.allclose(S[t + 1], cur, rtol=1e-12, atol=1e-300) or not np.isfinite(cur).all()


def test_ou_and_action_path():
    x = O.ou_step(np.array([0.0, 0.3]), np.array([1.0, -2.0]), 0.4, 0.6, 0.15, 1e-2)
    assert np.allclose(x, [0.0 + 0.15 * 0.4 * 0.01 + 0.6 * 0.1 * 1.0, 0.3 + 0.15 * 0.1 * 0.01 - 0.6 * 0.1 * 2.0])
    a = O.ddpg_action(np.array([0.9, -0.2]), np.array([0.5, 0.1]), 1.0)
    assert np.allclose(a, [1.0, -0.1])
    assert np.allclose(O.scale_action(np.array([-2.0, 0.0, 0.5]), -2.0, 2.0), [-2.0, 0.0, 1.0])
    g = O.ou_gaussian(7, np.arange(200000, dtype=np.uint64), 5)
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1) < 0.01
    # the four steps served by one Philox call (counter t >> 2: two word pairs x cos / sin output) are four
    # independent N(0,1) draws
    ids = np.arange(200000, dtype=np.uint64)
    gs = np.stack([O.ou_gaussian(7, ids, t) for t in (8, 9, 10, 11, 12)])
    assert np.all(np.abs(gs.mean(axis=1)) < 0.01) and np.all(np.abs(gs.std(axis=1) - 1) < 0.01)
    c = np.corrcoef(gs)
    assert np.max(np.abs(c - np.eye(5))) < 0.01
    assert np.max(np.abs(np.corrcoef(gs ** 2) - np.eye(5))) < 0.01


def test_pendulum_restatement_properties(oracle_clib):
    rng = np.random.default_rng(3)
    th, thd, a = rng.uniform(-10, 10, 1000), rng.uniform(-8, 8, 1000), rng.uniform(-3, 3, 1000)
    t0, d0, r0, u = O.pend_step(th, thd, a)
    t1, d1, r1, _ = O.pend_step(th, thd, a, v1_order=True)
    assert np.all(np.abs(u) <= 2) and np.all(np.abs(d0) <= 8) and np.all(r0 <= 0)
    unclipped = np.abs(d0) < 8
    assert np.allclose(t0[unclipped], t1[unclipped]) and np.allclose(r0, r1)
    # hanging straight down with zero torque is a fixed point of the angle dynamics' sign: gravity pulls to pi
    t, d, r, _ = O.pend_step(np.array([np.pi]), np.array([0.0]), np.array([0.0]))
    assert abs(d[0]) < 1e-14 and abs(r[0] + np.pi ** 2) < 1e-12
    # C twin
    ct, cd = th.copy(), thd.copy()
    cr = np.empty(1000)
    dp = ctypes.POINTER(ctypes.c_double)
    oracle_clib.ssc_oracle_pend_step(ctypes.c_int64(1000), ct.ctypes.data_as(dp), cd.ctypes.data_as(dp),
                                     a.ctypes.data_as(dp), ctypes.c_int(0), cr.ctypes.data_as(dp))
    assert np.allclose(ct, t0, atol=1e-13) and np.allclose(cd, d0, atol=1e-13) and np.allclose(cr, r0, atol=1e-12)
    obs = O.pend_obs(th, thd)
    assert obs.shape == (1000, 3) and np.allclose(obs[:, 0] ** 2 + obs[:, 1] ** 2, 1)


def test_kde_restatement_matches_scipy():
    """scipy.stats.gaussian_kde is what the reference calls (smartexplorationcontinuous.py:260,275)."""
    import scipy.stats
    rng = np.random.default_rng(0)
    for n, d in [(50, 2), (3000, 2), (500, 3), (200, 1)]:
        data = np.cumsum(rng.normal(size=(n, d)) * 0.02, axis=0)
        pts = np.concatenate([data[rng.integers(0, n, 40)], rng.normal(size=(10, d))])
        cov, wh, norm = O.kde_scott(data)
        k = scipy.stats.gaussian_kde(data.T, bw_method='scott')
        assert np.allclose(cov, k.covariance, rtol=1e-12)
        ref = k(pts.T)
        got = O.kde_evaluate(data, pts, wh, norm)
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-300)
    ucb, best = O.smart_start_ucb(np.array([0.0, 1.0, 0.5]), np.array([1.0, 1.0, 1e-6]), 1000, 0.01)
    assert best == 2 and np.isclose(ucb[0], np.sqrt(2 * np.log(1000) / (1000 * 0.01)))
    assert np.isclose(O.hyperellipsoid_volume([2.0, 3.0]), np.pi * 6)


def test_ddpg_train_step_matches_torch_autograd():
    """The manual backprop of oracle.ddpg_train_step against torch autograd + torch.optim-free Adam."""
    rng = np.random.default_rng(7)
    for llt in (True, False):
        aw = {k: v.astype(np.float64) for k, v in actor_weights(2, 64, 32, seed=1, w3_scale=0.3).items()}
        cw = dict(W1=rng.normal(size=(2, 64)) * 0.3, b1=rng.normal(size=64) * 0.1, W2=rng.normal(size=(65, 32)) * 0.2,
                  b2=rng.normal(size=32) * 0.1, W3=rng.normal(size=(32, 1)) * 0.2, b3=rng.normal(size=1) * 0.1)
        taw = {k: v + 0.01 * rng.normal(size=v.shape) for k, v in aw.items()}
        tcw = {k: v + 0.01 * rng.normal(size=v.shape) for k, v in cw.items()}
        B = 64
        batch = (rng.uniform(-1, 0.5, (B, 2)), rng.uniform(-1, 1, (B, 1)), rng.normal(size=B), rng.random(B) < 0.1,
                 rng.uniform(-1, 0.5, (B, 2)))
        na, nc = O.flatten_params(aw).size, O.flatten_params(cw).size
        adam = dict(m_actor=np.zeros(na), v_actor=np.zeros(na), t_actor=0, m_critic=np.zeros(nc), v_critic=np.zeros(nc), t_critic=0)
        a2, c2, ta2, tc2, adam2, closs, aloss = O.ddpg_train_step(aw, cw, taw, tcw, adam, batch, last_layer_tanh=llt)

        T = lambda d: {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in d.items()}
        pa, pc = T(aw), T(cw)
        s, a, r, t, s2 = (torch.tensor(np.asarray(x, np.float64)) for x in batch)
        act_fn = torch.tanh if llt else torch.relu

        def actor(p, x):
            return torch.tanh(act_fn(torch.relu(x @ p["W1"] + p["b1"]) @ p["W2"] + p["b2"]) @ p["W3"] + p["b3"])

        def critic(p, x, u):
            h = torch.cat([torch.relu(x @ p["W1"] + p["b1"]), u], dim=1)
            return act_fn(h @ p["W2"] + p["b2"]) @ p["W3"] + p["b3"]
        with torch.no_grad():
            y = r[:, None] + (1 - t[:, None]) * 0.99 * critic(T(tcw), s2, actor(T(taw), s2))
        closs_t = ((critic(pc, s, a) - y) ** 2).mean()
        gc = torch.autograd.grad(closs_t, [pc[k] for k in O.ACTOR_KEYS])
        aloss_t = -critic(pc, s, actor(pa, s)).mean()
        ga = torch.autograd.grad(aloss_t, [pa[k] for k in O.ACTOR_KEYS])
        assert abs(closs - closs_t.item()) < 1e-12 and abs(aloss - aloss_t.item()) < 1e-12

        def adam1(p, g, lr):      # first Adam step: m = (1-b1) g, v = (1-b2) g^2
            a_ = lr * np.sqrt(1 - 0.999) / (1 - 0.9)
            return p - a_ * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
        for i, k in enumerate(O.ACTOR_KEYS):
            assert np.allclose(c2[k], adam1(cw[k], gc[i].numpy(), 1e-3), rtol=1e-9, atol=1e-12), k
            assert np.allclose(a2[k], adam1(aw[k], ga[i].numpy(), 1e-4), rtol=1e-9, atol=1e-12), k
            assert np.allclose(ta2[k], 0.999 * taw[k] + 0.001 * a2[k]) and np.allclose(tc2[k], 0.999 * tcw[k] + 0.001 * c2[k])
        assert adam2["t_actor"] == 1 and adam2["t_critic"] == 1


def ln_params(rng, in_dim, h1, h2, out, extra=0):
    """Random actor / critic parameters WITH LayerNorm (gamma around 1, beta around 0: not the initial 1 / 0, so that both
    enter the gradients non-trivially); ``extra``: the critic's action rows of W2."""
    return dict(W1=rng.normal(size=(in_dim, h1)) * 0.5, b1=rng.normal(size=h1) * 0.1, ln1_b=rng.normal(size=h1) * 0.1,
                ln1_g=1.0 + 0.2 * rng.normal(size=h1), W2=rng.normal(size=(h1 + extra, h2)) * 0.3, b2=rng.normal(size=h2) * 0.1,
                ln2_b=rng.normal(size=h2) * 0.1, ln2_g=1.0 + 0.2 * rng.normal(size=h2), W3=rng.normal(size=(h2, out)) * 0.3,
                b3=rng.normal(size=out) * 0.1)


def test_layer_norm_ddpg_train_step_matches_torch_autograd():
    """layer_norm=True (models_editted.py:45-46,50-51,85-86,91-92; the class default): the oracle's manual LayerNorm
    forward / backward inside ddpg_train_step against torch autograd through torch.nn.functional.layer_norm with
    tc.layers.layer_norm's epsilon (1e-12), parameters in TF order (kernel, bias, beta, gamma per layer)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(11)
    for llt in (True, False):
        aw, cw = ln_params(rng, 3, 24, 12, 1), ln_params(rng, 3, 20, 16, 1, extra=1)
        taw = {k: v + 0.01 * rng.normal(size=v.shape) for k, v in aw.items()}
        tcw = {k: v + 0.01 * rng.normal(size=v.shape) for k, v in cw.items()}
        B = 40
        batch = (rng.normal(size=(B, 3)), rng.uniform(-1, 1, (B, 1)), rng.normal(size=B), rng.random(B) < 0.1, rng.normal(size=(B, 3)))
        assert O.param_keys(aw) == O.LN_KEYS and O.flatten_params(aw).size == 3 * 24 + 24 * 3 + 24 * 12 + 12 * 3 + 12 + 1
        na, nc = O.flatten_params(aw).size, O.flatten_params(cw).size
        adam = dict(m_actor=np.zeros(na), v_actor=np.zeros(na), t_actor=0, m_critic=np.zeros(nc), v_critic=np.zeros(nc), t_critic=0)
        a2, c2, ta2, tc2, adam2, closs, aloss = O.ddpg_train_step(aw, cw, taw, tcw, adam, batch, last_layer_tanh=llt)
        T = lambda d: {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in d.items()}
        pa, pc = T(aw), T(cw)
        s, a, r, t, s2 = (torch.tensor(np.asarray(x, np.float64)) for x in batch)
        act_fn = torch.tanh if llt else torch.relu
        ln = lambda x, p, w: F.layer_norm(x, (x.shape[-1],), p[w + "_g"], p[w + "_b"], eps=1e-12)

        def actor(p, x):
            h = torch.relu(ln(x @ p["W1"] + p["b1"], p, "ln1"))
            return torch.tanh(act_fn(ln(h @ p["W2"] + p["b2"], p, "ln2")) @ p["W3"] + p["b3"])

        def critic(p, x, u):
            h = torch.cat([torch.relu(ln(x @ p["W1"] + p["b1"], p, "ln1")), u], dim=1)
            return act_fn(ln(h @ p["W2"] + p["b2"], p, "ln2")) @ p["W3"] + p["b3"]
        # forward functions agree with the oracle's public ones
        lnp = lambda p: ((p["ln1_g"], p["ln1_b"]), (p["ln2_g"], p["ln2_b"]))
        core = lambda p: {k: p[k] for k in O.ACTOR_KEYS}
        assert np.allclose(O.actor_forward(batch[0], **core(aw), last_layer_tanh=llt, layer_norm=lnp(aw)), actor(pa, s).detach().numpy(), atol=1e-12)
        assert np.allclose(O.critic_forward(batch[0], batch[1], **core(cw), last_layer_tanh=llt, layer_norm=lnp(cw)),
                           critic(pc, s, a).detach().numpy(), atol=1e-12)
        with torch.no_grad():
            y = r[:, None] + (1 - t[:, None]) * 0.99 * critic(T(tcw), s2, actor(T(taw), s2))
        closs_t = ((critic(pc, s, a) - y) ** 2).mean()
        gc = torch.autograd.grad(closs_t, [pc[k] for k in O.LN_KEYS])
        aloss_t = -critic(pc, s, actor(pa, s)).mean()
        ga = torch.autograd.grad(aloss_t, [pa[k] for k in O.LN_KEYS])
        assert abs(closs - closs_t.item()) < 1e-12 and abs(aloss - aloss_t.item()) < 1e-12

        def adam1(p, g, lr):
            a_ = lr * np.sqrt(1 - 0.999) / (1 - 0.9)
            return p - a_ * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
        for i, k in enumerate(O.LN_KEYS):
            gck, gak = gc[i].numpy(), ga[i].numpy()
            # (Adam's first step is ~ -lr * sign(g): compare where the gradient is not within rounding of zero)
            okc, oka = np.abs(gck) > 1e-10, np.abs(gak) > 1e-10
            assert np.allclose(c2[k][okc], adam1(cw[k], gck, 1e-3)[okc], rtol=1e-7, atol=1e-12), k
            assert np.allclose(a2[k][oka], adam1(aw[k], gak, 1e-4)[oka], rtol=1e-7, atol=1e-12), k
            assert np.allclose(ta2[k], 0.999 * taw[k] + 0.001 * a2[k]) and np.allclose(tc2[k], 0.999 * tcw[k] + 0.001 * c2[k])
        # the gradients themselves, through the Adam moments (m = 0.1 g after the first step)
        gflat_c = np.concatenate([g.numpy().reshape(-1) for g in gc])
        gflat_a = np.concatenate([g.numpy().reshape(-1) for g in ga])
        assert np.allclose(adam2["m_critic"], 0.1 * gflat_c, rtol=1e-9, atol=1e-14)
        assert np.allclose(adam2["m_actor"], 0.1 * gflat_a, rtol=1e-9, atol=1e-14)


def test_ddpg_l2_regularisation_and_gradient_clipping_match_torch_autograd():
    """critic_l2_reg (ddpg_editted.py:183-191: scale * l2_loss over the critic's three dense kernels) and clip_norm
    (:175, :197: tf.clip_by_norm per variable) in oracle.ddpg_train_step against torch autograd, with and without LayerNorm."""
    rng = np.random.default_rng(5)
    l2, clip = 0.03, 0.3
    for ln in (False, True):
        if ln:
            aw, cw = ln_params(rng, 2, 16, 12, 1), ln_params(rng, 2, 20, 8, 1, extra=1)
        else:
            aw = {k: v.astype(np.float64) for k, v in actor_weights(2, 16, 12, seed=3, w3_scale=0.3).items()}
            cw = dict(W1=rng.normal(size=(2, 20)) * 0.5, b1=rng.normal(size=20) * 0.1, W2=rng.normal(size=(21, 8)) * 0.4,
                      b2=rng.normal(size=8) * 0.1, W3=rng.normal(size=(8, 1)) * 0.5, b3=rng.normal(size=1) * 0.1)
        keys = O.param_keys(aw)
        B = 32
        batch = (rng.normal(size=(B, 2)), rng.uniform(-1, 1, (B, 1)), rng.normal(size=B) * 3, rng.random(B) < 0.1, rng.normal(size=(B, 2)))
        na, nc = O.flatten_params(aw).size, O.flatten_params(cw).size
        adam = dict(m_actor=np.zeros(na), v_actor=np.zeros(na), t_actor=0, m_critic=np.zeros(nc), v_critic=np.zeros(nc), t_critic=0)
        a2, c2, _, _, adam2, closs, aloss = O.ddpg_train_step(aw, cw, aw, cw, adam, batch, critic_l2_reg=l2, clip_norm=clip)
        import torch.nn.functional as F
        T = lambda d: {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in d.items()}
        pa, pc = T(aw), T(cw)
        s, a, r, t, s2 = (torch.tensor(np.asarray(x, np.float64)) for x in batch)
        lnf = (lambda x, p, w: F.layer_norm(x, (x.shape[-1],), p[w + "_g"], p[w + "_b"], eps=1e-12)) if ln else (lambda x, p, w: x)

        def actor(p, x):
            h = torch.relu(lnf(x @ p["W1"] + p["b1"], p, "ln1"))
            return torch.tanh(torch.tanh(lnf(h @ p["W2"] + p["b2"], p, "ln2")) @ p["W3"] + p["b3"])

        def critic(p, x, u):
            h = torch.cat([torch.relu(lnf(x @ p["W1"] + p["b1"], p, "ln1")), u], dim=1)
            return torch.tanh(lnf(h @ p["W2"] + p["b2"], p, "ln2")) @ p["W3"] + p["b3"]
        with torch.no_grad():
            y = r[:, None] + (1 - t[:, None]) * 0.99 * critic(T(cw), s2, actor(T(aw), s2))
        closs_t = ((critic(pc, s, a) - y) ** 2).mean() + l2 * sum((pc[k] ** 2).sum() / 2 for k in ("W1", "W2", "W3"))
        gc = torch.autograd.grad(closs_t, [pc[k] for k in keys])
        ga = torch.autograd.grad(-critic(pc, s, actor(pa, s)).mean(), [pa[k] for k in keys])
        assert abs(closs - closs_t.item()) < 1e-12
        clipv = lambda g: g * (clip / max(float(g.norm()), clip))
        gflat_c = np.concatenate([clipv(g).numpy().reshape(-1) for g in gc])
        gflat_a = np.concatenate([clipv(g).numpy().reshape(-1) for g in ga])
        norms = sorted(float(g.norm()) for g in list(gc) + list(ga))
        assert norms[0] < clip < norms[-1], norms             # the clip binds for some variables only
        assert np.allclose(adam2["m_critic"], 0.1 * gflat_c, rtol=1e-9, atol=1e-14)
        assert np.allclose(adam2["m_actor"], 0.1 * gflat_a, rtol=1e-9, atol=1e-14)


def test_mlp_train_step_matches_torch_autograd_and_adam():
    """oracle.mlp_train_step (manual backprop + tf-style Adam) against torch autograd + torch.optim.Adam
    (same update rule as tf.train.AdamOptimizer up to where epsilon enters -- compared after ONE step where
    both reduce to -lr * sign-like update, and gradients compared exactly)."""
    rng = np.random.default_rng(3)
    dims = (3, 32, 16, 2)
    Ws = [rng.normal(size=(dims[i], dims[i + 1])) * 0.3 for i in range(3)]
    bs = [rng.normal(size=dims[i + 1]) * 0.1 for i in range(3)]
    x, z = rng.normal(size=(50, 3)), rng.normal(size=(50, 2))
    zero = lambda: dict(mW=[np.zeros_like(w) for w in Ws], vW=[np.zeros_like(w) for w in Ws],
                        mb=[np.zeros_like(b) for b in bs], vb=[np.zeros_like(b) for b in bs], t=0)
    nW, nb, adam, loss = O.mlp_train_step(Ws, bs, zero(), x, z, lr=1e-3)
    tW = [torch.tensor(w, requires_grad=True) for w in Ws]
    tb = [torch.tensor(b, requires_grad=True) for b in bs]
    h = torch.tensor(x)
    for l in range(3):
        h = h @ tW[l] + tb[l]
        if l < 2:
            h = torch.relu(h)
    tl = ((torch.tensor(z) - h) ** 2).mean()
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12
    for l in range(3):
        assert np.allclose(adam["mW"][l], 0.1 * tW[l].grad.numpy(), atol=1e-14)       # m = (1-b1) g
        assert np.allclose(adam["vb"][l], 0.001 * tb[l].grad.numpy() ** 2, atol=1e-16)
        g = tW[l].grad.numpy()
        lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
        assert np.allclose(nW[l], Ws[l] - lr_t * 0.1 * g / (np.sqrt(0.001 * g * g) + 1e-8), atol=1e-14)
    # batch composition: every old row is visited once per epoch, new rows are drawn with replacement
    batches = list(O.dyn_train_batches(1000, 300, 512, 0.9, 2, np.random.RandomState(0)))
    assert len(batches) == 2 * (1000 // (512 - 300)) and all(len(o) == 212 and len(n) == 300 for o, n in batches)
    per_epoch = np.concatenate([o for o, _ in batches[:4]])
    assert len(set(per_epoch.tolist())) == len(per_epoch)
    only_new = list(O.dyn_train_batches(10, 2000, 512, 1.0, 1, np.random.RandomState(0)))
    assert len(only_new) == 2000 // 512 and all(len(o) == 0 and len(n) == 512 for o, n in only_new)
