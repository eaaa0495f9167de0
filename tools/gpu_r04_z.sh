#!/bin/bash
O=gpurun_out/r04z; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_navigator.py tests/test_gpu_navigator_runs.py tests/test_gpu_smartstart_vec.py tests/test_gpu_smartstart_curves.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/prof_smartstart_vec.py 40 > $O/kt.log 2>&1 < /dev/null
python3 - $O/kt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:5]:
        print(r["Name"][:60], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
rm -rf $O/kt
step c4pe 300 python bench.py --config 4 --per-env-only --no-cpu-baseline > $O/c4pe.json 2> $O/c4pe.err < /dev/null
grep -o '"ms_per_step": [0-9.]*' $O/c4pe.json | head -1
