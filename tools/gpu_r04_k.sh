#!/bin/bash
# round 4, call k: the pair kernel of the small forward simulation (two rows per lane, packed FMAs)
O=gpurun_out/r04k; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 1000 python -m pytest tests/test_gpu_navigator.py tests/test_gpu_smartstart_vec.py tests/test_gpu_agents.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step c5 300 python bench.py --config 5 --steps 20 --warmup 5 > $O/c5.json 2> $O/c5.err < /dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04k/c5.json").read().strip().splitlines()[-1])
print({k:(v["ms_per_mpc_step"], v["sim_kernel_ms"]) for k,v in d["by_candidates"].items()})
PY
step ssvec 300 python tools/prof_smartstart_vec.py 40 > $O/ssvec.txt 2>&1 < /dev/null
tail -2 $O/ssvec.txt
