#!/bin/bash
# round 4, call H: LayerNorm learner cases, overlapped SmartStart selection (parity + loop timing)
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_h; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step tests 900 bash -c "python3 -m pytest tests/test_gpu_layer_norm.py tests/test_gpu_smartstart_vec.py tests/test_gpu_vec_learning.py -m gpu -q > $O/tests.log 2>&1; tail -25 $O/tests.log"
step ssloop 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 2>&1 | grep -v 'amdgpu.ids\|Warning\|publish' | tail -3 | tee $O/smartstart_vec_65536_overlap.txt"
step ssloop_seq 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 --sequential-selection 2>&1 | grep -v 'amdgpu.ids\|Warning\|publish' | tail -3 | tee $O/smartstart_vec_65536_sequential.txt"
