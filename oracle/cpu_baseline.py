"""CPU baseline legs of ``bench.py`` -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (like everything in ``oracle/``).

SURVEY.md section 8(d): the reference's own Python never travels to the GPU box, so the CPU baseline is this
repository's scalar restatement of ``rlTrain`` (smartstart/reinforcementLearningCore/rlTrain.py:63-114) around
``Continuous_MountainCarEnv_Editted.step`` (smartstart/environments/continuous_mountain_car_editted.py:60-82)
with a random policy (NN_Dynamics_Model/policy_random.py:14-15), run as P independent single-env processes the way the
reference fans experiments / rollouts out over ``multiprocessing.Pool`` (smartstart/utilities/experimenter.py:85-89,
NN_Dynamics_Model/collect_samples_threaded.py:31-40), with P = the cores this process may use.  The numpy-vectorised
fp64 restatement at 65 536 envs is timed on 1 and on P cores as the "strong CPU" line.

Workers are top-level functions so that a ``spawn`` pool can import them (bench.py has initialised the GPU by the
time it gets here, so ``fork`` is not an option).
"""
from __future__ import annotations

import os
import time


def usable_cores():
    """Cores this process may actually run on: min(os.cpu_count(), affinity mask, cgroup cpu quota)."""
    p = os.cpu_count() or 1
    try:
        p = min(p, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    p = min(p, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    p = min(p, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, p)


def scalar_rltrain_worker(args):
    """One process = one env, the reference's execution model: scalar Python rlTrain loop, random policy.
    -> (env-steps done, seconds)."""
    budget_s, seed = args
    import numpy as np

    from oracle import ssc_oracle as O
    env = O.ScalarMountainCar(1.0, 999, seed=seed)
    rng = np.random.RandomState(seed)
    low, high = -1.0, 1.0

    def get_action(_obs):                      # Policy_Random.get_action (policy_random.py:14-15)
        return rng.uniform(low, high, (1,))
    t0 = time.perf_counter()
    total = 0
    while time.perf_counter() - t0 < budget_s:
        _, steps = O.rl_train_scalar(env, get_action, num_episodes=2, max_steps=1000)
        total += steps
    return total, time.perf_counter() - t0


def numpy_vectorised_worker(args):
    """The numpy fp64 oracle on n envs with the engine's RNG and auto-reset (bench.py's GPU workload, on one core).
    -> (env-steps done, seconds)."""
    budget_s, n, env_id0 = args
    import numpy as np

    from oracle import ssc_oracle as O
    ids = np.arange(env_id0, env_id0 + n, dtype=np.uint64)
    pos, vel = O.mc_reset_state(1234, ids, O.RESET_T0)
    pos, vel = pos.astype(np.float64), vel.astype(np.float64)
    el = np.zeros(n, np.int64)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget_s:
        a = O.random_policy_actions(1234, ids, k).astype(np.float64)
        pos, vel, _r, d = O.mc_step(pos, vel, a)
        el += 1
        d = O.time_limit(d, el, 999)
        if d.any():
            rp, _ = O.mc_reset_state(1234, ids, k)
            pos = np.where(d, rp, pos)
            vel = np.where(d, 0.0, vel)
            el = np.where(d, 0, el)
        k += 1
    return n * k, time.perf_counter() - t0


def c_scalar_worker(args):
    """The C fp64 restatement (oracle/ssc_oracle.c), 4096 envs x 256-step chunks on one core.  -> (steps, seconds)."""
    budget_s, env_id0 = args
    import ctypes

    import numpy as np

    from oracle import ssc_oracle as O
    root = os.path.dirname(os.path.abspath(__file__))
    lib = ctypes.CDLL(os.path.join(root, "_build", "libssc_oracle.so"))
    lib.ssc_oracle_mc_rollout_random.restype = ctypes.c_int64
    nn, K = 4096, 256
    p, v = O.mc_reset_state(1234, np.arange(env_id0, env_id0 + nn, dtype=np.uint64), O.RESET_T0)
    p, v = p.astype(np.float64), v.astype(np.float64)
    st = np.zeros(nn, np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    t0 = time.perf_counter()
    done_steps, step0 = 0, 0
    while time.perf_counter() - t0 < budget_s:
        done_steps += lib.ssc_oracle_mc_rollout_random(
            ctypes.c_int64(nn), ctypes.c_int32(K), p.ctypes.data_as(dp), v.ctypes.data_as(dp),
            st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_double(0.0015), ctypes.c_int32(999),
            ctypes.c_uint64(1234), ctypes.c_uint64(env_id0), ctypes.c_uint64(step0), None)
        step0 += K
    return int(done_steps), time.perf_counter() - t0


def _rate(results):
    """Sum of per-process rates (each process is timed over its own busy interval)."""
    return float(sum(s / t for s, t in results if t > 0))


def run(budget_s=24.0, n_envs=65536, pool_cores=None):
    """All CPU legs -> dict for bench.py's ``cpu_baseline``.  Wall time ~ budget_s + pool start-up."""
    import multiprocessing as mp
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    P = pool_cores or usable_cores()
    leg = budget_s / 4.0
    out = {"unit": "env-steps/s", "kind": "port", "cores": P, "os_cpu_count": os.cpu_count()}
    c_ok = True
    try:
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle")], stdout=subprocess.DEVNULL)
    except Exception as e:  # noqa: BLE001  (the C half is optional for the baseline)
        c_ok = False
        out["c_scalar_error"] = str(e)
    ctx = mp.get_context("spawn")
    with ctx.Pool(P) as pool:
        pool.map(time.sleep, [0.0] * P)                      # start every worker before anything is timed
        r = pool.map(scalar_rltrain_worker, [(leg, 1234 + i) for i in range(P)])
        out["value"] = _rate(r)
        out["per_core"] = out["value"] / P
        out["sample"] = ("scalar Python rlTrain loop (rlTrain.py:63-114 restated) + MountainCar step + random policy, "
                         "%d independent single-env processes (experimenter.py:85-89 style pool), %.1f s each, "
                         "%d env-steps in total" % (P, leg, sum(s for s, _ in r)))
        r1 = numpy_vectorised_worker((leg, n_envs, 0))
        out["numpy_vectorised_1core"] = _rate([r1])
        per = max(1024, n_envs // P)
        rP = pool.map(numpy_vectorised_worker, [(leg, per, i * per) for i in range(P)])
        out["numpy_vectorised_pool"] = _rate(rP)
        out["numpy_vectorised_note"] = ("numpy fp64 oracle, engine RNG, auto-reset: %d envs on 1 core; %d x %d envs on "
                                        "%d processes" % (n_envs, P, per, P))
        if c_ok:
            try:
                rc1 = c_scalar_worker((leg / 2, 0))
                out["c_scalar_1core"] = _rate([rc1])
                rcP = pool.map(c_scalar_worker, [(leg / 2, i * 4096) for i in range(P)])
                out["c_scalar_pool"] = _rate(rcP)
            except Exception as e:  # noqa: BLE001
                out["c_scalar_error"] = str(e)
    return out


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print(json.dumps(run(float(sys.argv[1]) if len(sys.argv) > 1 else 8.0), indent=1))
