#!/usr/bin/env python3
"""gpurun_out/prof_r04_final/{c4_per_env,c4_h20,mpc_ref,ssvec,vec_ddpg} (tools/gpu_r04_final.sh) -> profiles/r04_final/<same>:
kernel-stats CSVs, per-kernel PMC means, the derived figures bench.py reads back (mfma_busy.json of the walking kernel,
valu_busy.json of the small-network simulation) and, for the actor-learner loop, how much of a chunk is kernels and how
much is launch boundaries.

    python3 tools/summarize_r04.py gpurun_out/prof_r04_final profiles/r04_final
"""
import collections, csv, glob, hashlib, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_SE = 1024, 32


def sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "smartstartcontinuous_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_of(d):
    pmc = {}
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            if k.startswith("void at::") or "elementwise" in k:
                continue
            pmc.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
    return pmc


def trace_of(d):
    for f in glob.glob(os.path.join(d, "kt", "*", "*_kernel_trace.csv")):
        return sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    return []


def main(src, dst):
    for name in ("c4_per_env", "c4_h20", "mpc_ref", "ssvec", "vec_ddpg"):
        s, d = os.path.join(src, name), os.path.join(dst, name)
        if not os.path.isdir(s):
            continue
        os.makedirs(d, exist_ok=True)
        for f in glob.glob(os.path.join(s, "kt", "*", "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(d, "kernel_stats.csv"))
        for f in ("bench.json", "loop.json", "out.txt"):
            fp = os.path.join(s, f)
            if os.path.exists(fp) and os.path.getsize(fp):
                open(os.path.join(d, f), "w").write("".join(l for l in open(fp) if "amdgpu.ids" not in l and "rocprofv3" not in l))
        pmc = pmc_of(s)
        if pmc:
            json.dump(pmc, open(os.path.join(d, "pmc_per_kernel.json"), "w"), indent=1)
        rows = trace_of(s)
        if rows:      # settled quarter of the dominant kernel
            by = collections.defaultdict(list)
            for r in rows:
                by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            k, dd = max(by.items(), key=lambda kv: sum(kv[1]))
            tail = sorted(dd[len(dd) * 3 // 4:])
            json.dump({"kernel": k, "launches": len(dd), "mean_us_all": sum(dd) / len(dd),
                       "last_quarter": {"n": len(tail), "mean_us": sum(tail) / len(tail), "median_us": tail[len(tail) // 2], "min_us": tail[0]}},
                      open(os.path.join(d, "kernel_trace_settled.json"), "w"), indent=1)
        if name in ("c4_per_env", "c4_h20"):
            names = [n for n in pmc if "dyn_mfma_sim_kernel" in n and (", true>" in n) == (name == "c4_per_env")]
            if names:
                c = pmc[names[0]]
                g = lambda key: c.get(key, {}).get("mean")
                busy, total = g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES")
                out = {"bound": "mfma_pipe", "kernel": names[0], "frac": (busy / N_SIMD) / (total / N_SE) if busy and total else None,
                       "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": total, "SQ_INSTS_MFMA": g("SQ_INSTS_MFMA"),
                       "note": "matrix-pipe busy cycles per SIMD over the launch's own length in shader cycles (same PMC pass)",
                       "source_sha": sha(("dyn_mfma.hip",))}
                json.dump(out, open(os.path.join(d, "mfma_busy.json"), "w"), indent=1)
                print(name, json.dumps(out))
        if name == "mpc_ref":
            names = [n for n in pmc if "dyn_small_sim" in n]
            if names:
                c = pmc[names[0]]
                g = lambda key: c.get(key, {}).get("mean")
                wave, valu, busy = g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU"), g("SQ_BUSY_CYCLES")
                # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count in units of four cycles (a wave64 VALU instruction occupies its SIMD
                # for four), SQ_BUSY_CYCLES in cycles per shader engine: the share of the launch in which a SIMD's vector ALU is busy
                out = {"bound": "valu", "kernel": names[0],
                       "frac": (valu * 4.0 / N_SIMD) / (busy / N_SE) if valu and busy else None,
                       "waves_resident_per_simd": (wave * 4.0 / N_SIMD) / (busy / N_SE) if wave and busy else None,
                       "SQ_ACTIVE_INST_VALU": valu, "SQ_WAVE_CYCLES": wave, "SQ_BUSY_CYCLES": busy, "SQ_INSTS_VALU": g("SQ_INSTS_VALU"),
                       "SQ_ACTIVE_INST_ANY": g("SQ_ACTIVE_INST_ANY"), "SQ_WAIT_INST_ANY": g("SQ_WAIT_INST_ANY"),
                       "note": "vector-ALU busy cycles per SIMD over the launch's own length in shader cycles (same PMC pass; per-SE sums)",
                       "source_sha": sha(("dyn_model.hip",))}
                json.dump(out, open(os.path.join(d, "valu_busy.json"), "w"), indent=1)
                print(name, json.dumps(out))
        if name == "vec_ddpg" and rows:
            # the steady half of the trace: kernel time vs wall time between the first and the last kernel
            half = rows[len(rows) // 2:]
            t0, t1 = int(half[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in half)
            busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in half)
            by = collections.defaultdict(lambda: [0, 0.0])
            for r in half:
                by[r["Kernel_Name"][:70]][0] += 1
                by[r["Kernel_Name"][:70]][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            out = {"launches": len(half), "wall_us": (t1 - t0) / 1e3, "kernel_us": busy / 1e3, "kernel_share": busy / (t1 - t0),
                   "gap_us_per_launch": ((t1 - t0) - busy) / 1e3 / len(half),
                   "by_kernel_us": {k: {"launches": v[0], "total_us": v[1]} for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]},
                   "note": "under rocprofv3 (kernel-trace): launches of one stream in order, the share of the second half's span in which a kernel ran"}
            json.dump(out, open(os.path.join(d, "launch_gaps.json"), "w"), indent=1)
            print(name, json.dumps({k: out[k] for k in ("launches", "wall_us", "kernel_us", "kernel_share", "gap_us_per_launch")}))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
