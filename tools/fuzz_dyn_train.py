"""Randomised sweep of the dynamics-model training step (forward, MSE, backprop, tf-style Adam) against the fp64 oracle:
random input / output widths, 1-3 hidden layers, depths on and off the 32- and power-of-two paddings, batches that do not
fill the last row block -- the fused one-launch kernels (one hidden layer) and the fp32-MFMA GEMM chain (deeper nets).
Development tool: python tools/fuzz_dyn_train.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import ssc_oracle as O
from smartstartcontinuous_amd import navigator as nav

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 404)
worst = 0.0
for case in range(cases):
    d, a = int(rng.integers(1, 9)), int(rng.integers(1, 5))
    layers = int(rng.choice([1, 1, 2, 3]))
    depth = int(rng.choice([1, 7, 20, 31, 32, 33, 64, 100, 127, 128, 200, 500, 512]))
    dims = (d + a,) + tuple(int(depth if l == 0 else rng.choice([depth, 16, 24, 40])) for l in range(layers)) + (d,)
    B = int(rng.choice([1, 31, 32, 33, 64, 77, 100, 512]))
    Ws = [(rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.normal(size=dims[i + 1]) * np.sqrt(2.0 / (1 + dims[i + 1]))).astype(np.float32) for i in range(len(dims) - 1)]
    norm = dict(mean_x=np.zeros(d), std_x=np.ones(d), mean_y=np.zeros(a), std_y=np.ones(a), mean_z=np.zeros(d), std_z=np.ones(d))
    model = nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=a)
    n = 1500
    X = rng.normal(size=(n, dims[0])).astype(np.float32)
    Z = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    Xd, Zd = torch.as_tensor(X, device="cuda"), torch.as_tensor(Z, device="cuda")
    oW, ob = [w.astype(np.float64) for w in Ws], [b.astype(np.float64) for b in bs]
    adam = dict(mW=[np.zeros_like(w) for w in oW], vW=[np.zeros_like(w) for w in oW], mb=[np.zeros_like(b) for b in ob],
                vb=[np.zeros_like(b) for b in ob], t=0)
    loss = torch.zeros(1, device="cuda")
    for step in range(3):
        idx = rng.permutation(n)[:B].astype(np.int32)
        oW, ob, adam, ref_loss = O.mlp_train_step(oW, ob, adam, X[idx], Z[idx], lr=1e-3)
        model.train_step(Xd, Zd, torch.as_tensor(idx, device="cuda"), lr=1e-3, loss=loss)
        assert abs(loss.item() - ref_loss) <= 2e-4 * max(1.0, ref_loss), (case, dims, B, step, loss.item(), ref_loss)
    err = max(max(float(np.max(np.abs(model.W[l].cpu().numpy() - oW[l]))), float(np.max(np.abs(model.b[l].cpu().numpy() - ob[l]))))
              for l in range(len(oW)))
    assert err <= 2e-5, (case, dims, B, err)
    worst = max(worst, err)
    print("case %2d dims %-26s batch %3d: max parameter deviation %.1e" % (case, dims, B, err), flush=True)
print("dynamics-model training step: %d random shapes ok, worst parameter deviation %.1e after 3 Adam steps of 1e-3" % (cases, worst))
