"""Does the placement of the transition log in memory matter?  (drift study: the two alternating chunk buffers of
bench.py differ by 4 % in median launch time.)  Times the BASELINE rollout into ONE big buffer with the chunk at
different base offsets and row paddings (the log ABI takes any row stride >= the packed row)."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smartstartcontinuous_amd import RandomPolicy, VecEnv, _ffi

n, K = 65536, 1024
env = VecEnv("MountainCarContinuous-v0", n, seed=1234)
env.reset()
pd, _ = env.policy_desc(RandomPolicy())
big = torch.empty((6 * n + 4096) * K + (64 << 20) // 4, dtype=torch.float32, device="cuda")
dbig = torch.empty((n + 4096) * K + (64 << 20), dtype=torch.uint8, device="cuda")
st = _ffi.RolloutState(env.s0.data_ptr(), env.s1.data_ptr(), env.steps.data_ptr(), env.ep_ret.data_ptr(), env.ou_x.data_ptr())

def make_log(base_off_bytes, pad_floats, dpad):
    log = _ffi.TransitionLog()
    b = big.data_ptr() + base_off_bytes
    for c in range(2):
        log.obs[c] = b + 4 * c * n
        log.obs2[c] = b + 4 * (4 + c) * n
    log.act = b + 4 * 2 * n
    log.rew = b + 4 * 3 * n
    log.done = dbig.data_ptr() + base_off_bytes // 4
    log.row_stride = 6 * n + pad_floats
    log.done_row_stride = n + dpad
    return log

def run(log, reps):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    t = 0
    for a, b in evs:
        a.record()
        _ffi.check(env.lib.ssc_rollout(ctypes.byref(env.params), ctypes.byref(pd), n, K, ctypes.byref(st), ctypes.byref(log), None,
                                       _ffi.ptr(env.stats), 1234, 0, t, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        b.record()
        t += K
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in evs]) * 1e3

run(make_log(0, 0, 0), 400)      # settle
cases = [(o, p, dp) for o in (0, 256, 4096, 65536, 1 << 20, 3 << 19, (1 << 21) + 4096) for p, dp in ((0, 0), (64, 0), (256, 64), (1024, 256), (4096, 1024))]
res = {c: [] for c in cases}
for rnd in range(3):               # interleaved rounds in one process
    for c in cases:
        res[c].append(run(make_log(*c), 120))
for c in cases:
    d = np.concatenate(res[c])
    print("base +%8d B  row pad %5d floats  done pad %5d : med %.1f us  min %.1f  p90 %.1f   (%.2f TB/s at median)" %
          (c[0], c[1], c[2], np.median(d), d.min(), np.percentile(d, 90), 25.0 * n * K / np.median(d) / 1e6), flush=True)
