"""Time VecEnv.rollout(K, MpcPolicy): P envs each following a recorded path with the navigator."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smartstartcontinuous_amd as ssc
from smartstartcontinuous_amd import navigator as nav
from smartstartcontinuous_amd.agents import init_dynamics_weights

def run(P, N, H, depth, K=400):
    rng = np.random.default_rng(0)
    Ws, bs = init_dynamics_weights(3, 2, 2 if depth >= 100 else 1, depth, torch.Generator().manual_seed(1))
    norm = dict(mean_x=[-0.5, 0.0], std_x=[0.2, 0.02], mean_y=[0.0], std_y=[0.6], mean_z=[0.0, 0.0], std_z=[0.01, 0.002])
    model = nav.DynamicsModel(Ws, bs, norm, 2, 1, precision="bf16_mfma")
    paths = [np.cumsum(rng.normal(scale=[0.01, 0.002], size=(80, 2)), axis=0) + [-0.5, 0.0] for _ in range(P)]
    env = ssc.VecEnv("MountainCarContinuous-v0", P, seed=3)
    env.reset()
    from smartstartcontinuous_amd import numerical as num
    wps, lefts, radii = [], [], []
    for pth in paths:
        stds, means = num.path_deltas_stds_and_means_per_dim(pth)
        r = num.radii_calc(means, stds, 1, 1, 1)
        wps.append(pth); radii.append(r); lefts.append(num.distances_left(pth, num.elliptical_euclidean_distance_function_generator(r)))
    ps = nav.MpcProblemSet(wps, lefts, radii, [0] * P)
    batch = nav.NavigatorBatch(model, ps, num_control_samples=N, horizon=H, seed=5)
    pol = ssc.MpcPolicy(batch)
    env.rollout(K, pol)     # same K as the timed call: the captured graph (and its log chunk) is per chunk length
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.rollout(K, pol)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps(dict(P=P, N=N, H=H, depth=depth, ms_per_step=dt * 1e3, env_steps_per_s=P / dt)), flush=True)

if __name__ == "__main__":
    run(16, 4096, 4, 500)
    run(1, 5000, 4, 32)
    run(256, 256, 4, 500)
