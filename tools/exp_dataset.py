"""Timing of the data-collection path at BASELINE size: 65 536 random rollouts (one env each) -> training set.
Prints per-kernel time and the HBM rate on the algorithmic bytes (read chunk columns + write rows)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smartstartcontinuous_amd as ssc  # noqa: E402
from smartstartcontinuous_amd import _ffi, collect_samples as cs  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for env_id, K in (("Pendulum-v0", 200), ("MountainCarContinuous-v0", 333)):
    n = 65536
    env = ssc.VecEnv(env_id, n, device="cuda", seed=1)
    chunk = env.rollout(K, policy=ssc.RandomPolicy())
    d = chunk.obs_dim
    lib = _ffi.lib()
    lens = torch.empty(n, dtype=torch.int32, device="cuda")
    off = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    log = chunk.as_struct()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(int(lib.ssc_dataset_scan_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    t_scan = timed(lambda: _ffi.check(lib.ssc_dataset_scan(ctypes.byref(log), K, n, _ffi.ptr(lens), _ffi.ptr(off),
                                                            _ffi.ptr(ws), ws.numel(), st)))
    rows = int(off[n].item())
    X = torch.empty((rows, d), device="cuda"); Y = torch.empty((rows, 1), device="cuda"); Z = torch.empty((rows, d), device="cuda")
    t_build = timed(lambda: _ffi.check(lib.ssc_dataset_build(ctypes.byref(log), d, K, n, _ffi.ptr(lens), _ffi.ptr(off), rows,
                                                             _ffi.ptr(X), _ffi.ptr(Y), _ffi.ptr(Z), st)))
    t_stats = timed(lambda: cs.column_stats(X))
    inputs = torch.empty((rows, d + 1), device="cuda")
    mean, std = cs.column_stats(X)
    t_z = timed(lambda: cs.zscore_into(X, mean, std, inputs, 0))
    mean_y, std_y = cs.column_stats(Y)
    t_zc = timed(lambda: cs.zscore_concat(X, mean, std, Y, mean_y, std_y, out=inputs))
    t_noise = timed(lambda: cs.add_noise_device(X, 0.01, 1, 0, mean=mean.abs()))
    t_roll = timed(lambda: env.rollout(K, policy=ssc.RandomPolicy(), out=chunk), reps=5)
    build_bytes = rows * ((3 * d + 1) * 4 + (2 * d + 1) * 4)       # read obs, obs2, act; write X, Y, Z
    print(f"{env_id}: {n} rollouts x {K} steps -> {rows} rows | rollout {t_roll:.3f} ms | scan {t_scan:.3f} ms "
          f"({n * K / t_scan / 1e6:.1f} GB/s of done bytes) | build {t_build:.3f} ms ({build_bytes / t_build / 1e6:.0f} GB/s) | "
          f"column_stats {t_stats:.3f} ms ({2 * rows * d * 4 / t_stats / 1e6:.0f} GB/s) | zscore {t_z:.3f} ms "
          f"({2 * rows * d * 4 / t_z / 1e6:.0f} GB/s) | zscore_concat {t_zc:.3f} ms ({2 * rows * (d + 1) * 4 / t_zc / 1e6:.0f} GB/s) | add_noise {t_noise:.3f} ms ({2 * rows * d * 4 / t_noise / 1e6:.0f} GB/s)")
