"""The small forward simulation alone at the reference's MPC shape (N = 5000 x 209 problems, H = 4, NND_MB 1x32): launch time, and
under `rocprofv3 --pmc ...` the counters of dyn_small_sim*_kernel.  argv: [launches] [N] [H]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from smartstartcontinuous_amd import navigator as nav
from smartstartcontinuous_amd.agents import init_dynamics_weights
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4
d, a = 2, 1
P = max(1, (1 << 20) // N); M = P * N
norm = dict(mean_x=np.array([-0.5, 0.0]), std_x=np.array([0.3, 0.03]), mean_y=np.array([0.0]), std_y=np.array([0.58]),
            mean_z=np.array([0.0, 0.0]), std_z=np.array([0.01, 0.002]))
Ws, bs = init_dynamics_weights(d + a, d, 1, 32, torch.Generator().manual_seed(1234))
model = nav.DynamicsModel(Ws, bs, norm, d, a, precision="f32")
s0 = (torch.rand((P, d), device="cuda") - 0.5) * torch.tensor([1.0, 0.1], device="cuda")
S = torch.empty((H + 1, M, d), device="cuda")
def run(t):
    model.do_forward_sim_sampled(s0, nav.mpc_sampling(N, [-1.0], [1.0], 1234, 0, t), M, H, out=S)
for t in range(20): run(t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for t in range(launches): run(t)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / launches
print(json.dumps({"rows": M, "N": N, "H": H, "us_per_launch": round(us, 2), "row_steps_per_s": M * H / us * 1e6}))
