"""Time Dyn_Model.train on the device (DynamicsModel.train -> ssc_mlp_train_step) for the reference's data-set size
(8 300 rows, batch 512, 30 epochs) and a few network shapes."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smartstartcontinuous_amd import navigator as nav
from smartstartcontinuous_amd.agents import init_dynamics_weights
rng = np.random.default_rng(0)
n = 8300
X = rng.normal(size=(n, 3)); Z = rng.normal(size=(n, 2)) * 0.1 + X[:, :2] * 0.05
norm = dict(mean_x=[0, 0], std_x=[1, 1], mean_y=[0], std_y=[1], mean_z=[0, 0], std_z=[1, 1])
for layers, depth in [(1, 32), (1, 500), (2, 500)]:
    Ws, bs = init_dynamics_weights(3, 2, layers, depth, torch.Generator().manual_seed(1))
    model = nav.DynamicsModel(Ws, bs, norm, 2, 1, precision="f32")
    model.train(X, Z, np.zeros((0, 3)), np.zeros((0, 2)), 1, 0.0, rng=np.random.RandomState(0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = model.train(X, Z, np.zeros((0, 3)), np.zeros((0, 2)), 30, 0.0, rng=np.random.RandomState(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = 30 * (n // 512)
    print(json.dumps(dict(layers=layers, depth=depth, seconds=dt, steps=steps, ms_per_step=dt / steps * 1e3, last_loss=loss)), flush=True)

# GPU time per step without the host-side epoch bookkeeping: 2000 steps enqueued by ONE ssc_mlp_train_steps call
for layers, depth, B in [(1, 32, 512), (1, 500, 512), (1, 500, 64), (2, 500, 512)]:
    Ws, bs = init_dynamics_weights(3, 2, layers, depth, torch.Generator().manual_seed(1))
    model = nav.DynamicsModel(Ws, bs, norm, 2, 1, precision="f32")
    Xd = torch.as_tensor(X, dtype=torch.float32, device="cuda"); Zd = torch.as_tensor(Z, dtype=torch.float32, device="cuda")
    steps = 2000 if layers == 1 else 200
    idx = torch.randint(0, n, (steps, B), dtype=torch.int32, device="cuda")
    model.train_steps(Xd, Zd, idx[:10])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); model.train_steps(Xd, Zd, idx); e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(json.dumps(dict(layers=layers, depth=depth, batch=B, steps=steps, gpu_us_per_step=e0.elapsed_time(e1) / steps * 1e3,
                          host_enqueue_us_per_step=(t1 - t0) / steps * 1e6)), flush=True)
