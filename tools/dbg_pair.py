import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import ssc_oracle as O
from smartstartcontinuous_amd import navigator as nav
from tests.test_gpu_navigator import make_mlp, make_norm
rng = np.random.default_rng(5)
P, N, H, seed = 3, 400, 4, 31
Ws, bs = make_mlp(rng, (3, 32, 2))
nm = make_norm(rng, 2, 1)
model = nav.DynamicsModel(Ws, bs, nm, state_dim=2, act_dim=1, precision="f32")
s0 = torch.as_tensor(rng.normal(size=(P, 2)) * 0.3, dtype=torch.float32, device="cuda")
nm32 = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in nm.items()}
for with_A in (False, True):
    for t in (0, 1, 5):
        sp = nav.mpc_sampling(N, [-1.0], [1.0], seed, 0, t)
        A_out = torch.zeros((P * N, H, 1), device="cuda") if with_A else None
        S = model.do_forward_sim_sampled(s0, sp, P * N, H, A_out=A_out).cpu().numpy()
        for p in range(P):
            A = O.mpc_action_samples(seed, p, N, H, 1, t, [-1.0], [1.0])
            ref = O.dyn_forward_sim(s0[p].cpu().numpy(), A, nm32, Ws, bs)
            got = S[:, p * N:(p + 1) * N]
            err = np.abs(got - ref).max(axis=(0, 2))
            bad = np.where(err > 1e-4)[0]
            print("with_A", with_A, "t", t, "p", p, "max err", err.max(), "bad rows", bad[:10], len(bad))
            if with_A:
                print("   A err", np.abs(A_out.cpu().numpy()[p * N:(p + 1) * N] - A).max())
