#!/bin/bash
# round 4, call F: fused-actor SmartStart step + fast compaction: parity, step timing with kernel stats, loop timing
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_f; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step tests 600 bash -c "python3 -m pytest tests/test_gpu_smartstart_vec.py tests/test_gpu_smartstart_curves.py tests/test_gpu_vec_learning.py -m gpu -q > $O/tests.log 2>&1; tail -8 $O/tests.log"
step ssvec_kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ssvec_kt -- python3 tools/prof_smartstart_vec.py 40 > $O/ssvec_kt.log 2>&1
f=$(ls $O/ssvec_kt/*/*_kernel_stats.csv | head -1); head -6 $f | cut -c1-160
grep -v amdgpu.ids $O/ssvec_kt.log | tail -1 | cut -c1-400
step ssloop 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 2>&1 | grep -v 'amdgpu.ids\|Warning\|publish' | tail -3 | tee $O/smartstart_vec_65536.txt"
step loop 200 bash -c "python3 tools/exp_pipeline.py wide 2>&1 | grep -v amdgpu.ids | head -1 | tee $O/vec_ddpg_loop.txt"
