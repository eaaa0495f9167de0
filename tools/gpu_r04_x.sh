#!/bin/bash
O=gpurun_out/r04x; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 1100 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_learning_curves.py > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/prof_smartstart_vec.py 40 > $O/kt.log 2>&1 < /dev/null
python3 - $O/kt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:60], "avg ns", r["AverageNs"], "min", r["MinNs"], "calls", r["Calls"])
PY
rm -rf $O/kt
step ssvec 300 python tools/prof_smartstart_vec.py 40 > $O/ssvec.txt 2>&1 < /dev/null
tail -1 $O/ssvec.txt | grep -o '"ms_per_step": [0-9.]*'
