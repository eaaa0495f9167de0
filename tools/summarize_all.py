#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/{c2,c3,c4} (tools/profile_all.sh) -> profiles/<name>/{c2,c3,c4}: kernel stats CSV,
per-kernel PMC means, bench line, and for c2 the HBM traffic check."""
import collections, csv, glob, json, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
for cfg in ("c2", "c3", "c4", "train", "dataset"):
    s, d = os.path.join(src, cfg), os.path.join(dst, cfg)
    if not os.path.isdir(s):
        continue
    os.makedirs(d, exist_ok=True)
    for f in glob.glob(os.path.join(s, "kt", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(d, "kernel_stats.csv"))
    # settled launches of the dominant kernel from the full trace (kernel_stats.csv averages every launch of the run,
    # the post-idle transient and the settling phase included)
    for f in glob.glob(os.path.join(s, "kt", "*", "*_kernel_trace.csv")):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        by = collections.defaultdict(list)
        for r in rows:
            by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        name, dd = max(by.items(), key=lambda kv: sum(kv[1]))
        tail = sorted(dd[len(dd) * 3 // 4:])
        json.dump({"kernel": name, "launches": len(dd), "mean_us_all": sum(dd) / len(dd),
                   "last_quarter": {"n": len(tail), "mean_us": sum(tail) / len(tail), "median_us": tail[len(tail) // 2],
                                    "min_us": tail[0], "p90_us": tail[int(0.9 * len(tail))]}},
                  open(os.path.join(d, "kernel_trace_settled.json"), "w"), indent=1)
    pmc = {}
    for f in glob.glob(os.path.join(s, "pmc_*", "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            if k.startswith("void at::") or "elementwise" in k:
                continue
            pmc.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
    json.dump(pmc, open(os.path.join(d, "pmc_per_kernel.json"), "w"), indent=1)
    bj = os.path.join(s, "bench.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        shutil.copy(bj, os.path.join(d, "bench.json"))
    ot = os.path.join(s, "out.txt")
    if os.path.exists(ot):        # tool output measured under the profiler (kernels serialised): kept for the kernel list
        open(os.path.join(d, "tool_output_under_profiler.txt"), "w").write(
            "".join(l for l in open(ot) if "amdgpu.ids" not in l))
    if cfg == "c2":
        k = [v for name, v in pmc.items() if "rollout_kernel" in name]
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        if k:
            w = k[0].get("WRITE_SIZE", {}).get("mean", 0) * 1024
            r = k[0].get("FETCH_SIZE", {}).get("mean", 0) * 1024 * 2
            alg = b["roofline"]["algorithmic_bytes_per_launch"]
            # stamped with the sha of the kernel sources the box ran (= this tree: summarise right after the gpurun call
            # that measured), so that bench.py can tell a stale figure from a current one
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            json.dump({"write_bytes": w, "fetch_bytes_corrected": r, "algorithmic_bytes": alg,
                       "traffic_over_algorithmic": (w + r) / alg, "source_sha": bench.source_sha()},
                      open(os.path.join(dst, "traffic.json"), "w"), indent=1)
            print("c2 traffic/algorithmic = %.4f" % ((w + r) / alg))
    st = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(st):
        print(cfg, open(st).read().splitlines()[1][:200])
