"""Column placement variant of exp_align.py: the fp32 columns of a step sit N floats apart in the packed chunk, i.e. exact
multiples of 256 KB at N = 65 536 -- does padding BETWEEN the columns (every column pointer of the log ABI is free) change
the write rate?  (original docstring follows)
Does the placement of the transition log in memory matter?  (drift study: the two alternating chunk buffers of
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
big = torch.empty((6 * (n + 40960) + 4096) * K + (64 << 20) // 4, dtype=torch.float32, device="cuda")
dbig = torch.empty((n + 4096) * K + (64 << 20), dtype=torch.uint8, device="cuda")
st = _ffi.RolloutState(env.s0.data_ptr(), env.s1.data_ptr(), env.steps.data_ptr(), env.ep_ret.data_ptr(), env.ou_x.data_ptr())

def make_log(cpad, order):
    """columns cpad floats apart beyond N; order = the slot of (obs0, obs1, act, rew, obs2_0, obs2_1) inside a row"""
    log = _ffi.TransitionLog()
    b = big.data_ptr()
    cs = n + cpad
    slot = dict(zip(("o0", "o1", "a", "r", "p0", "p1"), order))
    log.obs[0], log.obs[1] = b + 4 * slot["o0"] * cs, b + 4 * slot["o1"] * cs
    log.act, log.rew = b + 4 * slot["a"] * cs, b + 4 * slot["r"] * cs
    log.obs2[0], log.obs2[1] = b + 4 * slot["p0"] * cs, b + 4 * slot["p1"] * cs
    log.done = dbig.data_ptr()
    log.row_stride = 6 * cs
    log.done_row_stride = n
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

packed = (0, 1, 2, 3, 4, 5)
run(make_log(0, packed), 600)      # settle
cases = [(p, packed) for p in (int(x) for x in (sys.argv[1:] or "0 16 64 256 1024 2048 4096".split()))]
res = {c: [] for c in cases}
for rnd in range(3):               # interleaved rounds in one process
    for c in cases:
        res[c].append(run(make_log(*c), 120))
for c in cases:
    d = np.concatenate(res[c])
    print("column pad %5d floats (%6d B): med %.1f us  min %.1f  p90 %.1f   (%.2f TB/s at median)" %
          (c[0], 4 * c[0], np.median(d), d.min(), np.percentile(d, 90), 25.0 * n * K / np.median(d) / 1e6), flush=True)
