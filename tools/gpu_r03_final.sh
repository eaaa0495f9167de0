#!/bin/bash
# Round-3 closing run on the GPU box: full GPU suite, smoke, the vectorised SmartStart example at two sizes, then the
# profile set (bench lines, rocprofv3 kernel-trace stats, SEPARATE PMC passes) for the three measured configurations.
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_final; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 900 bash -c "python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
step smoke 200 bash -c "python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; tail -1 $O/smoke.log"
step ssvec_small 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 4096 --chunks 40 --samples 64 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smartstart_vec_4096.txt"
step ssvec_big 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smartstart_vec_65536.txt"
bash tools/profile_all.sh r03_final
bash tools/profile_pmc.sh 3 gpurun_out/prof_r03_final/pmc_c3
