#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 1000 python -m pytest tests/test_gpu_smartstart_vec.py tests/test_gpu_navigator.py tests/test_gpu_agents.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step ssvec 300 python tools/prof_smartstart_vec.py 40 > $O/ssvec.txt 2>&1 < /dev/null
tail -1 $O/ssvec.txt | grep -o '"ms_per_step": [0-9.]*'
step c5 300 python bench.py --config 5 --steps 20 --warmup 5 > $O/c5.json 2> $O/c5.err < /dev/null
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04p/c5.json").read().strip().splitlines()[-1])
print({k:(v["ms_per_mpc_step"], v["sim_kernel_ms"]) for k,v in d["by_candidates"].items()})
PY
step c5kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --config 5 --steps 20 --warmup 5 > $O/kt.log 2>&1 < /dev/null
f=$(ls $O/kt/*/*_kernel_stats.csv | head -1); [ -n "$f" ] && cut -c1-140 "$f" | head -5 | tee $O/c5_kernel_stats_head.txt
rm -rf $O/kt
step c4pe 300 python bench.py --config 4 --per-env-only --no-cpu-baseline > $O/c4pe.json 2> $O/c4pe.err < /dev/null
grep -o '"ms_per_step": [0-9.]*' $O/c4pe.json | head -1
