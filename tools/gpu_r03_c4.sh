#!/bin/bash
# config-4 profile without the per-env leg + the vectorised SmartStart example on the MFMA forward simulation
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03_final; O=gpurun_out/r03_final; mkdir -p $OUT/c4 $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES"
rm -rf $OUT/c4/kt $OUT/c4/pmc_sq
step c4_kt 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4/kt -- python3 bench.py --config 4 --no-per-env --no-cpu-baseline --no-single-step --no-other-configs > $OUT/c4/kt.log 2>&1
step c4_pmc 120 rocprofv3 --pmc $SQ --output-format csv -d $OUT/c4/pmc_sq -- python3 bench.py --config 4 --no-per-env --no-cpu-baseline --no-single-step --no-other-configs --steps 5 --warmup 2 --steady-launches 0 --settle-launches 0 > $OUT/c4/pmc_sq.log 2>&1
step ssvec_small 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 4096 --chunks 40 --samples 64 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smartstart_vec_4096.txt"
step ssvec_big 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smartstart_vec_65536.txt"
