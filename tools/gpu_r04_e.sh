#!/bin/bash
# round 4, call E: compaction parity + the vectorised SmartStart step / loop timings + the reference-shape MPC leg
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_e; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step tests 600 bash -c "python3 -m pytest tests/test_gpu_smartstart_vec.py tests/test_gpu_navigator.py tests/test_gpu_agents.py tests/test_gpu_vec_learning.py -m gpu -q > $O/tests.log 2>&1; tail -8 $O/tests.log"
step ssvec 300 bash -c "python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/smartstart_vec_leg.json
import argparse, json, sys, torch
sys.path.insert(0, '.')
import bench
a = argparse.Namespace(steps=20, warmup=5, settle_launches=300)
r = bench.bench_smartstart_vec(a, torch, emit=False)
print(json.dumps({k: r[k] for k in ('value', 'ms_per_step', 'gpu_ms_per_step', 'navigated_fraction_last_chunk')}))
PY"
step c5 300 bash -c "python3 bench.py --config 5 --settle-launches 300 2>/dev/null | tee $O/c5.json | cut -c1-900"
step ssloop 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smartstart_vec_65536.txt"
