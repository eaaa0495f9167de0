#!/bin/bash
# rollout parity tests + the config-2 / config-3 bench lines (after a change to csrc/rollout.hip)
set -u
export TMPDIR=/tmp
O=gpurun_out/rollout_ab
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 400 bash -c "python3 -m pytest tests/test_gpu_env.py tests/test_gpu_actor_pendulum.py tests/test_gpu_dataset.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
step c2 200 bash -c "python3 bench.py --no-cpu-baseline --no-single-step --no-other-configs 2>/dev/null | python3 -c \"import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('c2', d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_dist']['median'], r['steady']['median'], r['steady']['min'])\""
step c3 200 bash -c "python3 bench.py --config 3 --no-cpu-baseline 2>/dev/null | python3 -c \"import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('c3', d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_dist']['median'])\""
