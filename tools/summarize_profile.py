#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile_bench.sh) into the small files
committed under profiles/: kernel stats CSV, per-kernel PMC averages, and the bench line."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
# settled launches of the dominant kernel from the full trace (kernel_stats.csv averages every launch of the run, the
# post-idle transient and the settling phase included)
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_trace.csv")):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    name, d = max(by.items(), key=lambda kv: sum(kv[1]))
    tail = sorted(d[len(d) * 3 // 4:])
    json.dump({"kernel": name, "launches": len(d), "mean_us_all": sum(d) / len(d),
               "last_quarter": {"n": len(tail), "mean_us": sum(tail) / len(tail), "median_us": tail[len(tail) // 2],
                                "min_us": tail[0], "p90_us": tail[int(0.9 * len(tail))]}},
              open(os.path.join(dst, "kernel_trace_settled.json"), "w"), indent=1)
pmc = {}
for tagdir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(tagdir):
        continue
    for f in glob.glob(os.path.join(tagdir, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            pmc.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
json.dump(pmc, open(os.path.join(dst, "pmc_per_kernel.json"), "w"), indent=1)
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, "bench.json"))
    b = json.loads(open(bj).read().strip().splitlines()[-1])
    k = [v for name, v in pmc.items() if "rollout_kernel" in name]
    if k and "algorithmic_bytes_per_launch" in b["roofline"] and "WRITE_SIZE" in k[0]:
        w = k[0].get("WRITE_SIZE", {}).get("mean", 0) * 1024
        r = k[0].get("FETCH_SIZE", {}).get("mean", 0) * 1024 * 2   # gfx950: FETCH_SIZE reads 1/2 (MI355X_MICROARCH.md HBM)
        alg = b["roofline"]["algorithmic_bytes_per_launch"]
        print("rollout kernel: WRITE_SIZE %.4g B, FETCH_SIZE(x2) %.4g B, algorithmic %.4g B, traffic/alg = %.4f"
              % (w, r, alg, (w + r) / alg))
        # stamped with the sha of the kernel sources the box ran (= this tree: summarise right after the gpurun
        # call that measured), so bench.py can tell a stale figure from a current one
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        json.dump({"write_bytes": w, "fetch_bytes_corrected": r, "algorithmic_bytes": alg,
                   "traffic_over_algorithmic": (w + r) / alg, "source_sha": bench.source_sha()},
                  open(os.path.join(dst, "traffic.json"), "w"), indent=1)
for extra in ("out.txt",):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, "tool_output_under_profiler.txt"))
print(open(os.path.join(dst, "kernel_stats.csv")).read()[:600])
