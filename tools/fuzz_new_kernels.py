"""Randomised differential check of the training and data-set kernels against the fp64 oracle over many shapes
(run on the GPU box; prints the worst deviations).  Not part of the test suite -- the fixed cases in tests/ are."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ssc_oracle as O  # noqa: E402
import smartstartcontinuous_amd as ssc  # noqa: E402
from smartstartcontinuous_amd import collect_samples as cs, navigator as nav  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


def make_mlp(dims):
    Ws = [rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1])) for i in range(len(dims) - 1)]
    bs = [rng.normal(size=dims[i + 1]) * np.sqrt(2.0 / (1 + dims[i + 1])) for i in range(len(dims) - 1)]
    return [w.astype(np.float32) for w in Ws], [b.astype(np.float32) for b in bs]


worst = 0.0
for trial in range(40):
    L = int(rng.integers(1, 4))
    d_out = int(rng.integers(1, 9))
    a = int(rng.integers(1, 5))
    dims = (d_out + a,) + tuple(int(rng.integers(1, 130)) if rng.random() < 0.7 else int(rng.integers(130, 513)) for _ in range(L)) + (d_out,)
    B = int(rng.choice([1, 2, 31, 32, 33, 64, 77, 128, 500, 512]))
    Ws, bs = make_mlp(dims)
    norm = dict(mean_x=np.zeros(d_out), std_x=np.ones(d_out), mean_y=np.zeros(a), std_y=np.ones(a), mean_z=np.zeros(d_out),
                std_z=np.ones(d_out))
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d_out, act_dim=a)
    n = 700
    X = rng.normal(size=(n, dims[0])).astype(np.float32)
    Z = (rng.normal(size=(n, dims[-1])) * 0.5).astype(np.float32)
    Xd, Zd = torch.as_tensor(X, device="cuda"), torch.as_tensor(Z, device="cuda")
    oW, ob = [w.astype(np.float64) for w in Ws], [b.astype(np.float64) for b in bs]
    adam = dict(mW=[np.zeros_like(w) for w in oW], vW=[np.zeros_like(w) for w in oW],
                mb=[np.zeros_like(b) for b in ob], vb=[np.zeros_like(b) for b in ob], t=0)
    idx = np.stack([rng.permutation(n)[:B] for _ in range(3)]).astype(np.int32)
    losses = model.train_steps(Xd, Zd, torch.as_tensor(idx, device="cuda"), lr=1e-3).cpu().numpy()
    for k in range(3):
        oW, ob, adam, ref_loss = O.mlp_train_step(oW, ob, adam, X[idx[k]], Z[idx[k]], lr=1e-3)
        assert abs(losses[k] - ref_loss) <= 3e-4 * max(1.0, ref_loss), (dims, B, k, losses[k], ref_loss)
    dev = max(max(np.max(np.abs(model.W[l].cpu().numpy() - oW[l])) for l in range(len(oW))),
              max(np.max(np.abs(model.b[l].cpu().numpy() - ob[l])) for l in range(len(ob))))
    worst = max(worst, dev)
    # Adam's first steps have magnitude ~lr whatever the gradient: a flipped sign of a ~0 gradient shows up as 2e-3;
    # anything systematic (wrong tile, missing row) is far larger on at least one parameter per layer
    assert dev <= 2.5e-3, (dims, B, dev)
    frac_bad = np.mean([np.mean(np.abs(model.W[l].cpu().numpy() - oW[l]) > 5e-5) for l in range(len(oW))])
    assert frac_bad < 0.01, (dims, B, frac_bad)
print("training: 40 random shapes ok, worst parameter deviation %.2e" % worst)

for trial in range(30):
    K, n, d = int(rng.integers(1, 400)), int(rng.integers(1, 300)), int(rng.integers(1, 4))
    p = float(rng.choice([0.0, 0.003, 0.05, 0.5]))
    obs = rng.normal(size=(K, n, d)).astype(np.float32)
    act = rng.normal(size=(K, n, 1)).astype(np.float32)
    done = (rng.random((K, n)) < p).astype(np.uint8)
    obs2 = rng.normal(size=(K, n, d)).astype(np.float32)
    cont = done[:-1] == 0
    obs2[:-1][cont] = obs[1:][cont]
    ch = ssc.TransitionChunk(d, K, n, "cuda", packed=bool(rng.integers(0, 2)))
    ch.obs.copy_(torch.as_tensor(obs.transpose(2, 0, 1))); ch.obs2.copy_(torch.as_tensor(obs2.transpose(2, 0, 1)))
    ch.act.copy_(torch.as_tensor(act[:, :, 0])); ch.rew.zero_(); ch.done.copy_(torch.as_tensor(done))
    ts = cs.dataset_from_chunk(ch)
    st, ct = O.rollouts_from_chunk(obs, act, done)
    rows = sum(max(len(s) - 1, 0) for s in st)
    assert len(ts) == rows, (K, n, d, p)
    if rows:
        Xr, Yr = O.generate_training_data_inputs(st, ct)
        Zr = O.generate_training_data_outputs(st)
        assert np.array_equal(ts.dataX.cpu().numpy(), Xr) and np.array_equal(ts.dataY.cpu().numpy(), Yr)
        assert np.array_equal(ts.dataZ.cpu().numpy(), Zr), (K, n, d, p)
        m, s = cs.column_stats(ts.dataX)
        mr, sr = O.column_stats(Xr)
        assert np.allclose(m.cpu().numpy(), mr, rtol=1e-12, atol=1e-14) and np.allclose(s.cpu().numpy(), sr, rtol=1e-11, atol=1e-14)
print("data set: 30 random chunks ok")
