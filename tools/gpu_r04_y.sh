#!/bin/bash
O=gpurun_out/r04y; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 1000 python -m pytest tests/test_gpu_navigator.py tests/test_gpu_smartstart_vec.py tests/test_gpu_agents.py tests/test_gpu_smartstart_curves.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
for v in product stepold product stepold; do
  if [ $v = product ]; then unset SSC_LIB_PATH; else export SSC_LIB_PATH=$PWD/tools/_build/libssc_$v.so; fi
  step kt_$v 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$v -- python3 tools/prof_smartstart_vec.py 40 > $O/kt_$v.log 2>&1 < /dev/null
  python3 - $O/kt_$v $v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "mpc_rollout_step" in r["Kernel_Name"]]
    d.sort()
    q = lambda x: d[int(x * (len(d) - 1))]
    print(sys.argv[2], "step kernel us: n", len(d), "mean %.1f min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (sum(d) / len(d), d[0], q(.1), q(.5), q(.9), q(.99), d[-1]))
PY
  rm -rf $O/kt_$v
done
