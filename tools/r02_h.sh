#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02h
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest_nav 300 bash -c "python3 -m pytest tests/test_gpu_navigator.py -x -q -m gpu > $O/pytest_nav.log 2>&1; tail -8 $O/pytest_nav.log"
step pytest 300 bash -c "python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log"
step bench4 200 bash -c "python3 bench.py --config 4 --cpu-budget 4 --steps 200 --warmup 20 > $O/bench_c4.json 2> $O/bench_c4.err; tail -2 $O/bench_c4.err; python3 -c \"import json;d=json.load(open('$O/bench_c4.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'])\""
step kt4 300 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt4 -- python3 bench.py --config 4 --no-cpu-baseline --steps 200 --warmup 20 > $O/kt4.log 2>&1; python3 - <<'PY'
import csv,glob
for f in glob.glob('$O/kt4/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY"
