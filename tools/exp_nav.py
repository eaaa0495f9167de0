"""Experiment driver: forward-sim + MPC scoring timings (BASELINE config 4 shapes)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from smartstartcontinuous_amd import navigator as nav


def make(dims, seed=0):
    rng = np.random.default_rng(seed)
    Ws = [(rng.normal(size=(dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(rng.normal(size=dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(dims) - 1)]
    d, a = dims[-1], dims[0] - dims[-1]
    norm = dict(mean_x=np.zeros(d), std_x=np.ones(d), mean_y=np.zeros(a), std_y=np.ones(a), mean_z=np.zeros(d), std_z=np.full(d, 0.05))
    return nav.DynamicsModel(Ws, bs, norm, state_dim=d, act_dim=a), d, a


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def run(dims, M, H, prec):
    model, d, a = make(dims)
    A = torch.rand((M, H, a), device="cuda") * 2 - 1
    s0 = torch.randn((M, d), device="cuda") * 0.3
    S = torch.empty((H + 1, M, d), device="cuda")
    med, mn = timeit(lambda: model.do_forward_sim(s0, A, precision=prec, out=S))
    flop = 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    print(json.dumps(dict(dims=dims, M=M, H=H, prec=prec, ms_med=med, ms_min=mn, row_steps_per_s=M * H / med * 1e3,
                          TFLOPs=flop * M * H / med / 1e9)), flush=True)


if __name__ == "__main__":
    run((4, 500, 500, 3), 65536, 4, "bf16_mfma")
    run((4, 500, 500, 3), 65536, 20, "bf16_mfma")
    run((4, 500, 500, 3), 262144, 4, "bf16_mfma")
    run((4, 500, 500, 3), 65536, 4, "f32")
    run((3, 500, 2), 65536, 4, "bf16_mfma")
    run((3, 32, 2), 65536, 4, "bf16_mfma")
    run((3, 32, 2), 65536, 4, "f32")
    # MPC scoring: 16 problems x 4096 samples, H=4
    rng = np.random.default_rng(0)
    P, N, H, d = 16, 4096, 4, 2
    wps = [np.cumsum(rng.normal(scale=[0.02, 0.004], size=(200, d)), axis=0) + [-0.5, 0] for _ in range(P)]
    radii = [np.array([0.03, 0.006])] * P
    lefts = [np.cumsum(np.ones(200))[::-1].copy() for _ in range(P)]
    ps = nav.MpcProblemSet(wps, lefts, radii, [3] * P)
    S = torch.randn((H + 1, P * N, d), device="cuda") * 0.05 + torch.tensor([-0.5, 0.0], device="cuda")
    med, mn = timeit(lambda: nav.mpc_score(ps, S))
    print(json.dumps(dict(mpc_score=True, P=P, N=N, H=H, ms_med=med, ms_min=mn)))
