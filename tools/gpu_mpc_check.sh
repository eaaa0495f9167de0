#!/bin/bash
# navigator parity tests + the config-4 bench line + per-kernel times of the MPC step (after a change to csrc/mpc.hip / dyn_mfma.hip)
set -u
export TMPDIR=/tmp
O=gpurun_out/mpc_check
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 400 bash -c "python3 -m pytest tests/test_gpu_navigator.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
step kt4 300 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt4 -- python3 bench.py --config 4 --no-cpu-baseline --steps 200 --warmup 20 > $O/kt4.log 2>&1; python3 - <<'PY'
import csv,glob
for f in glob.glob('$O/kt4/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY"
step bench4 300 bash -c "python3 bench.py --config 4 --no-cpu-baseline 2>/dev/null | python3 -c \"import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c4', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])\""
