#!/bin/bash
O=gpurun_out/r04z; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_agents.py tests/test_gpu_smartstart_vec.py tests/test_gpu_multirank.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/prof_vec_ddpg_loop.py 100 > $O/kt.log 2>&1 < /dev/null
python3 - $O/kt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "replay_append" in r["Name"] or "rollout_kernel" in r["Name"] or "fixed_kernel" in r["Name"]:
            print(r["Name"][:60], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
rm -rf $O/kt
step kt2 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 12 --samples 16 --plans 8 > $O/kt2.log 2>&1 < /dev/null
python3 - $O/kt2 <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "replay_append" in r["Name"]:
            print(r["Name"][:60], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
rm -rf $O/kt2
