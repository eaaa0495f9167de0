#!/bin/bash
# round 4, call G: LayerNorm parity + regression of everything touched (learner, rollout, smartstart step)
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_g; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step ln 600 bash -c "python3 -m pytest tests/test_gpu_layer_norm.py -m gpu -q -x > $O/ln.log 2>&1; tail -30 $O/ln.log"
step regress 900 bash -c "python3 -m pytest tests/test_gpu_agents.py tests/test_gpu_actor_pendulum.py tests/test_gpu_smartstart_vec.py tests/test_gpu_env.py -m gpu -q > $O/regress.log 2>&1; tail -8 $O/regress.log"
step ssvec 300 bash -c "python3 tools/prof_smartstart_vec.py 40 2>&1 | grep -v amdgpu.ids | grep -o '\"ms_per_step\": [0-9.]*, \"gpu_ms_per_step\": [0-9.]*, \"navigated_fraction_last_chunk\": [0-9.]*' | tee $O/ssvec.txt"
