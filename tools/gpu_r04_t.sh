#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
for rep in 1 2 3; do
for v in product stepv; do
  if [ $v = product ]; then unset SSC_LIB_PATH; else export SSC_LIB_PATH=$PWD/tools/_build/libssc_$v.so; fi
  step kt_$v 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v$rep -- python3 tools/prof_smartstart_vec.py 40 > $O/kt_$v$rep.log 2>&1 < /dev/null
  python3 - $O/kt_$v$rep $v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "mpc_rollout_step" in r["Name"]:
            print(sys.argv[2], "step kernel avg ns", r["AverageNs"], "min", r["MinNs"], "calls", r["Calls"])
PY
  rm -rf $O/kt_$v$rep
done
done
