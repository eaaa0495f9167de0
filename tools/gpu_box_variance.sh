#!/bin/bash
# box-to-box variance: the three bench lines on whatever box this call got (tools/gpu_box_variance.sh <tag>)
set -u
export TMPDIR=/tmp
O=gpurun_out/var_$1
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step c2 240 bash -c "python3 bench.py --no-cpu-baseline --no-single-step --no-other-configs > $O/c2.json 2>/dev/null"
step c3 240 bash -c "python3 bench.py --config 3 --no-cpu-baseline > $O/c3.json 2>/dev/null"
step c4 240 bash -c "python3 bench.py --config 4 --no-cpu-baseline > $O/c4.json 2>/dev/null"
step train 120 bash -c "python3 tools/exp_train.py > $O/train.txt 2>/dev/null"
(rocm-smi --showpower --showclocks 2>/dev/null | head -30) > $O/smi.txt
python3 - <<PY
import json
row = {}
for c in ("c2", "c3", "c4"):
    d = json.loads(open("$O/%s.json" % c).read().strip().splitlines()[-1])
    row[c] = dict(value=d["value"], ms_per_step=d["ms_per_step"], kernel_ms=d["roofline"]["kernel_ms"], frac=d["roofline"]["frac"],
                  steady_median=d["roofline"].get("steady", {}).get("median"))
row["learner_us_per_iter"] = json.loads(open("$O/train.txt").read().strip().splitlines()[-2])["us_per_iter"]
print(json.dumps(row))
open("$O/row.json", "w").write(json.dumps(row))
PY
