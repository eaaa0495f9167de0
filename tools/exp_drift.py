"""Is the headline kernel's launch-to-launch drift (238 -> 332 us inside one 20-launch bench window, round 1)
clock / power management, and what is the steady state?

Runs on the GPU box.  Three launch patterns of the BASELINE configs[1] rollout (65 536 envs x 1024 steps, two
alternating chunk buffers, exactly bench.py's loop), every launch bracketed by its own pair of events on the
launch stream and NO host sync inside a series:

  continuous : N launches back to back (steady state under sustained load)
  bursts     : B bursts of 25 launches separated by an idle gap (what a default bench.py run looks like)
  membw      : the bare store loops of tools/membw.hip as the same kind of series (the "ceiling" as a distribution)

Beside them (a) a one-wave clock probe on a side stream samples (s_memrealtime, s_memtime) every ~50 us -> the
shader clock as a time series, and (b) a host thread samples amdsmi gpu metrics (gfx/mem/fabric clocks, socket
power, throttle status, HBM temperature) as fast as the driver answers.  Everything goes to one JSON file.

    python tools/exp_drift.py --out gpurun_out/drift.json [--launches 3000]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class SmiSampler(threading.Thread):
    """amdsmi gpu-metrics sampler (all visible devices; the busy one is picked afterwards by power)."""

    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.errors, self.stop_flag = [], [], False
        self.fields = None

    def run(self):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
        except Exception as e:  # noqa: BLE001
            self.errors.append("init: %r" % (e,))
            return
        want = ("current_gfxclk", "current_uclk", "current_socclk", "average_socket_power", "current_socket_power",
                "throttle_status", "temperature_hotspot", "temperature_mem", "average_gfx_activity",
                "average_umc_activity", "indep_throttle_status", "current_gfxclks", "accumulation_counter",
                "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "hbm_thm_residency_acc",
                "gfxclk_lock_status", "firmware_timestamp", "system_clock_counter", "gfx_below_host_limit_acc")
        while not self.stop_flag:
            for di, h in enumerate(handles):
                t = time.perf_counter()
                try:
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                except Exception as e:  # noqa: BLE001
                    if len(self.errors) < 5:
                        self.errors.append("metrics: %r" % (e,))
                    time.sleep(0.01)
                    continue
                row = {"t": t, "dev": di}
                for k in want:
                    v = m.get(k)
                    if isinstance(v, (list, tuple)):
                        v = [x for x in v if isinstance(x, (int, float)) and x not in (65535, 4294967295)][:8]
                    if isinstance(v, (int, float, list)):
                        row[k] = v
                self.samples.append(row)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/drift.json")
    ap.add_argument("--launches", type=int, default=3000)
    ap.add_argument("--bursts", type=int, default=20)
    ap.add_argument("--burst-len", type=int, default=25)
    ap.add_argument("--gap-ms", type=float, default=40.0)
    ap.add_argument("--no-probe", action="store_true")
    args = ap.parse_args()

    import torch

    from smartstartcontinuous_amd import RandomPolicy, TransitionChunk, VecEnv

    n, K = 65536, 1024
    env = VecEnv("MountainCarContinuous-v0", n, seed=1234)
    env.reset()
    chunks = [TransitionChunk(2, K, n, env.device) for _ in range(2)]
    pd = env.policy_desc(RandomPolicy())
    out = {"n": n, "K": K, "bytes_per_launch": 25.0 * n * K, "device": torch.cuda.get_device_name(0)}

    smi = SmiSampler()
    smi.start()

    probe = None
    if not args.no_probe:
        try:
            probe = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libclockprobe.so"))
            probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        except OSError as e:
            out["probe_error"] = str(e)
    side = torch.cuda.Stream()

    def run_series(name, pattern):
        """pattern: list of ('launch', count) / ('sleep', seconds)."""
        total = sum(c for kind, c in pattern if kind == "launch")
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(total)]
        for i in range(5):
            env.rollout(K, out=chunks[i & 1], policy_desc=pd)
        torch.cuda.synchronize()
        time.sleep(0.25)        # start every series from the same (idle) power state
        pbuf = None
        if probe is not None:
            est_s = total * 0.00035 + sum(c for kind, c in pattern if kind == "sleep") + 0.05
            ns = min(int(est_s / 50e-6) + 200, 1 << 18)
            pbuf = torch.zeros(2 * ns, dtype=torch.int64, device="cuda")
            with torch.cuda.stream(side):
                rc = probe.clock_probe_launch(ctypes.c_void_p(pbuf.data_ptr()), ns, 12, ctypes.c_void_p(side.cuda_stream))
            assert rc == 0
        t_host0 = time.perf_counter()
        i = 0
        for kind, c in pattern:
            if kind == "sleep":
                torch.cuda.current_stream().synchronize()
                time.sleep(c)
                continue
            for _ in range(c):
                a, b = evs[i]
                a.record()
                env.rollout(K, out=chunks[i & 1], policy_desc=pd)
                b.record()
                i += 1
        torch.cuda.synchronize()
        t_host1 = time.perf_counter()
        res = {"t_host0": t_host0, "t_host1": t_host1,
               "start_us": [round(evs[0][0].elapsed_time(a) * 1e3, 1) for a, _ in evs],
               "dur_us": [round(a.elapsed_time(b) * 1e3, 1) for a, b in evs]}
        if pbuf is not None:
            p = pbuf.cpu().numpy().reshape(-1, 2)
            p = p[p[:, 0] != 0]
            res["probe_realtime_100MHz"] = (p[:, 0] - p[0, 0]).tolist()
            res["probe_cycles"] = (p[:, 1] - p[0, 1]).tolist()
        out[name] = res
        d = sorted(res["dur_us"])
        print(name, "launches", len(d), "min %.1f med %.1f p90 %.1f max %.1f us" %
              (d[0], d[len(d) // 2], d[int(len(d) * 0.9)], d[-1]), flush=True)

    run_series("continuous", [("launch", args.launches)])
    bursts = []
    for _ in range(args.bursts):
        bursts += [("launch", args.burst_len), ("sleep", args.gap_ms * 1e-3)]
    run_series("bursts", bursts)
    # paced: the same launches with the host waiting for each one (a launch gap of ~20-40 us between kernels)
    run_series("continuous_2", [("launch", args.launches)])

    # the bare store loops, as a separate process while this one stays idle on the GPU (the sampler keeps running)
    mb = os.path.join(ROOT, "tools", "_build", "membw")
    if os.path.exists(mb):
        t0 = time.perf_counter()
        r = subprocess.run([mb, "series", str(min(args.launches, 2000))], capture_output=True, text=True, timeout=300)
        out["membw"] = {"t_host0": t0, "t_host1": time.perf_counter(), "rc": r.returncode, "series": []}
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                s = json.loads(line)
                out["membw"]["series"].append(s)
                d = sorted(s["dur_us"])
                gbs = s["bytes_per_launch"] / 1e3
                print("membw", s["shape"], "TB/s at min/med/p90 time: %.2f %.2f %.2f" %
                      (gbs / d[0] / 1e3, gbs / d[len(d) // 2] / 1e3, gbs / d[int(len(d) * 0.9)] / 1e3), flush=True)
        if r.returncode != 0:
            out["membw"]["stderr"] = r.stderr[-2000:]

    smi.stop_flag = True
    smi.join(timeout=2.0)
    out["smi"] = {"samples": smi.samples, "errors": smi.errors}
    print("smi samples", len(smi.samples), "errors", smi.errors[:2], flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
