#!/bin/bash
O=gpurun_out/r04u; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_navigator.py tests/test_gpu_smartstart_vec.py tests/test_gpu_agents.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step c5kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --config 5 --steps 20 --warmup 5 > $O/kt.log 2>&1 < /dev/null
python3 - $O/kt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ssc::" in r["Name"]:
            print(r["Name"][:60], "avg ns", r["AverageNs"], "calls", r["Calls"])
PY
rm -rf $O/kt
step c5 300 python bench.py --config 5 --steps 20 --warmup 5 > $O/c5.json 2> $O/c5.err < /dev/null
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04u/c5.json").read().strip().splitlines()[-1])
print({k:(v["ms_per_mpc_step"], v["sim_kernel_ms"]) for k,v in d["by_candidates"].items()})
PY
