#!/bin/bash
# kernel trace of the vectorised actor-learner loop (tools/exp_pipeline_hostprof.py): per-kernel totals and the
# per-chunk GPU timeline (busy time vs gaps)
set -u
export TMPDIR=/tmp
O=gpurun_out/prof_pipe
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python3 tools/exp_pipeline_hostprof.py > $O/log.txt 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/run_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
t = glob.glob("$O/**/run_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                       # past the warm-up call
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("launches", len(rows), "span_ms %.3f busy_ms(sum of kernels, streams overlap) %.3f" % (span / 1e6, busy / 1e6))
# one chunk's timeline
names = [r["Kernel_Name"][:50] for r in rows]
i0 = next(i for i, n in enumerate(names) if "rollout_kernel" in n and i > 40)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 14]:
    print("%8.1f us  +%7.1f us  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
PY
