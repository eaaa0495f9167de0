#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02r
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --force-dist --steps 20 --warmup 5 --settle-launches 200 --steady-launches 0 --no-cpu-baseline --no-single-step --no-other-configs > $O/kt.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r02r/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-120:]
prev_end = None
for r in rows[:60]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-60s q%s start+%8.1f us dur %7.1f us gap %6.1f' % (r['Kernel_Name'][:60], r.get('Queue_Id', '?'), (s - int(rows[0]['Start_Timestamp'])) / 1e3, (e - s) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
PY
