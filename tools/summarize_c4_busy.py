#!/usr/bin/env python3
"""Matrix-pipe busy fraction of the forward-simulation kernel (BASELINE configs[3]) from the committed PMC pass of
tools/profile_all.sh (profiles/<tag>/c4/pmc_per_kernel.json):

    python3 tools/summarize_c4_busy.py profiles/r03_final/c4

writes <dir>/mfma_busy.json = {frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines), the
counters, source_sha of dyn_mfma.hip}.  Both counters come from the SAME launches of the same pass, so the ratio does not
depend on the clock the chip held.  bench.py reads it into roofline.pipe_busy of the config-4 line while dyn_mfma.hip still
hashes to source_sha."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C4_SOURCES = ("dyn_mfma.hip",)            # bench.py's list
N_SIMD, N_SE = 1024, 32                   # MI355X: 256 CUs x 4 SIMDs; 8 XCDs x 4 shader engines


def source_sha():
    h = hashlib.sha256()
    for f in C4_SOURCES:
        h.update(open(os.path.join(ROOT, "smartstartcontinuous_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main(d):
    pmc = json.load(open(os.path.join(d, "pmc_per_kernel.json")))
    # the bench's kernel: in-kernel candidate sampling (last template argument 1) when it was profiled, else any
    # (round 4: the last template argument says whether the block walks over row tiles -- the per-env leg's kernel)
    want_walk = len(sys.argv) > 2 and sys.argv[2] == "walk"
    names = sorted((n for n in pmc if "dyn_mfma_sim_kernel" in n and (", true>" in n) == want_walk),
                   key=lambda n: (", 1, " not in n, n))
    if not names:
        print("no dyn_mfma_sim_kernel in", d)
        return 1
    c = pmc[names[0]]
    g = lambda k: c[k]["mean"] if isinstance(c.get(k), dict) else c.get(k)
    busy, total, n_mfma = g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES"), g("SQ_INSTS_MFMA")
    out = {"bound": "mfma_pipe", "kernel": names[0], "frac": (busy / N_SIMD) / (total / N_SE),
           "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES": total, "SQ_INSTS_MFMA": n_mfma,
           "busy_cycles_per_simd": busy / N_SIMD, "launch_cycles": total / N_SE,
           "note": "cycles in which a SIMD's matrix pipe was executing an MFMA (16 per v_mfma_f32_16x16x32_bf16) over the "
                   "launch's own length in shader cycles, both from the same PMC pass: independent of the clock held",
           "source_sha": source_sha()}
    json.dump(out, open(os.path.join(d, "mfma_busy.json"), "w"), indent=1)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
