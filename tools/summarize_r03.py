#!/usr/bin/env python3
"""The issue-cycle roof of the fused actor rollout (BASELINE configs[2]) from the PMC passes of tools/profile_pmc.sh:

    python3 tools/summarize_r03.py gpurun_out/r03_diag/pmc_c3 profiles/r03/c3

writes <dst>/valu_issue.json = {bound, frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, the counters it was computed from,
source_sha of the kernel sources} and <dst>/pmc_per_kernel.json (every counter of every pass, per kernel).  bench.py reads
valu_issue.json into roofline.valu_issue of the config-3 line when the kernel sources still hash to source_sha."""
import collections, csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C3_SOURCES = ("rollout.hip", "ssc_device.h", "actor_device.h")      # bench.py's list


def source_sha():
    h = hashlib.sha256()
    for f in C3_SOURCES:
        h.update(open(os.path.join(ROOT, "smartstartcontinuous_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    pmc = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            if k.startswith("void at::") or "elementwise" in k:
                continue
            pmc.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
    json.dump(pmc, open(os.path.join(dst, "pmc_per_kernel.json"), "w"), indent=1)
    k = [(name, v) for name, v in pmc.items() if "rollout_kernel" in name and "Actor" in name]
    if not k:
        print("no actor rollout kernel in", src)
        return 1
    name, c = k[0]
    g = lambda n: c.get(n, {}).get("mean")
    wave, valu = g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU")
    out = {"bound": "valu_issue", "kernel": name,
           "frac": valu / wave if wave and valu else None,
           "SQ_ACTIVE_INST_VALU": valu, "SQ_WAVE_CYCLES": wave,
           "SQ_INSTS_VALU": g("SQ_INSTS_VALU"), "SQ_INSTS_VALU_TRANS_F32": g("SQ_INSTS_VALU_TRANS_F32"),
           "SQ_INSTS_MFMA": g("SQ_INSTS_MFMA"), "SQ_VALU_MFMA_BUSY_CYCLES": g("SQ_VALU_MFMA_BUSY_CYCLES"),
           "SQ_WAIT_INST_ANY": g("SQ_WAIT_INST_ANY"), "SQ_ACTIVE_INST_ANY": g("SQ_ACTIVE_INST_ANY"),
           "note": "share of the resident wave's cycles (one wave per SIMD) in which a VALU instruction was issuing; the rest "
                   "is dependent-issue stall, store issue and MFMA wait.  Both counters are per-SE sums of the same launch, "
                   "so their ratio needs no unit correction.",
           "source_sha": source_sha()}
    json.dump(out, open(os.path.join(dst, "valu_issue.json"), "w"), indent=1)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
