#!/bin/bash
# Round-4 closing run, second pass (after the scorer changes): GPU suite, smoke, and the legs the scorer is part of --
# config 5, the vectorised SmartStart step and loop, the config-4 per-env leg.  Same layout as tools/gpu_r04_final.sh.
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_final; P=gpurun_out/prof_r04_final; mkdir -p $O $P
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES"
SQV="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES"
rm -rf $P/mpc_ref $P/ssvec $P/c4_per_env
mkdir -p $P/c4_per_env $P/mpc_ref $P/ssvec
step pytest 1100 bash -c "python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
step smoke 200 bash -c "python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; tail -1 $O/smoke.log"
step ssvec_big 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 2>&1 | grep -v 'amdgpu.ids\|RuntimeWarning\|self.pool.publish' | tail -3 | tee $O/smartstart_vec_65536.txt"
step ssvec_seq 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 --sequential-selection 2>&1 | grep -v 'amdgpu.ids\|RuntimeWarning\|self.pool.publish' | tail -3 | tee $O/smartstart_vec_65536_sequential.txt"
step c4pe_kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/c4_per_env/kt -- python3 bench.py --config 4 --per-env-only --no-cpu-baseline > $P/c4_per_env/bench.json 2> $P/c4_per_env/kt.err
step c4pe_pmc 200 rocprofv3 --pmc $SQ --output-format csv -d $P/c4_per_env/pmc_sq -- python3 bench.py --config 4 --per-env-only --no-cpu-baseline --steps 4 > $P/c4_per_env/pmc_sq.log 2>&1
step c5_bench 200 bash -c "python3 bench.py --config 5 > $P/mpc_ref/bench.json 2> $P/mpc_ref/bench.err"
step c5_kt 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/mpc_ref/kt -- python3 bench.py --config 5 > $P/mpc_ref/kt.log 2>&1
step c5_pmc 200 rocprofv3 --pmc $SQV --output-format csv -d $P/mpc_ref/pmc_sq -- python3 bench.py --config 5 --steps 5 --warmup 2 --settle-launches 0 > $P/mpc_ref/pmc_sq.log 2>&1
step ssvec_plain 200 bash -c "python3 tools/prof_smartstart_vec.py 40 2>&1 | tail -1 > $P/ssvec/unprofiled.txt; cut -c1-300 $P/ssvec/unprofiled.txt"
step ssvec_kt 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ssvec/kt -- python3 tools/prof_smartstart_vec.py 40 > $P/ssvec/out.txt 2>&1
step default_bench 300 bash -c "python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json"
