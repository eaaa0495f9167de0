#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/exp_pipeline_hostprof.py > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r02v/kt/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print('%-80s calls %6s avg %10.1f us total %8.2f ms' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
f = glob.glob('gpurun_out/r02v/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))[-40:]
t0 = int(rows[0]['Start_Timestamp']); pe = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-50s start+%8.1f dur %7.1f gap %7.1f' % (r['Kernel_Name'][:50], (s-t0)/1e3, (e-s)/1e3, 0 if pe is None else (s-pe)/1e3)); pe = e
PY
