#!/bin/bash
# Run on the GPU box (via gpurun): the bench line, rocprofv3 kernel-trace stats and SEPARATE PMC passes for the three
# measured configurations.  usage: tools/profile_all.sh <tag>  -> gpurun_out/prof_<tag>/{c2,c3,c4,...}
# Every step runs under its own timeout; a step killed at its limit ends the script (no further GPU work).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT/c2 $OUT/c3 $OUT/c4 $OUT/train $OUT/dataset
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
SQ="SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES"
# config 2 (headline)
step c2_bench 240 bash -c "python3 bench.py > $OUT/c2/bench.json 2> $OUT/c2/bench.err"
step c2_kt 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2/kt -- python3 bench.py --no-cpu-baseline --no-single-step --no-other-configs > $OUT/c2/kt.log 2>&1
step c2_pmc_w 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c2/pmc_write -- python3 bench.py --no-cpu-baseline --no-single-step --no-other-configs --steps 5 --warmup 2 --steady-launches 0 --settle-launches 0 > $OUT/c2/pmc_write.log 2>&1
step c2_pmc_f 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-single-step --no-other-configs --steps 5 --warmup 2 --steady-launches 0 --settle-launches 0 > $OUT/c2/pmc_fetch.log 2>&1
# config 3 (actor MFMA rollout)
step c3_bench 240 bash -c "python3 bench.py --config 3 --cpu-budget 8 > $OUT/c3/bench.json 2> $OUT/c3/bench.err"
step c3_kt 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3/kt -- python3 bench.py --config 3 --no-cpu-baseline --no-single-step --no-other-configs > $OUT/c3/kt.log 2>&1
step c3_pmc 120 rocprofv3 --pmc $SQ --output-format csv -d $OUT/c3/pmc_sq -- python3 bench.py --config 3 --no-cpu-baseline --no-single-step --no-other-configs --steps 5 --warmup 2 --steady-launches 0 --settle-launches 0 > $OUT/c3/pmc_sq.log 2>&1
# config 4 (dynamics MLP forward sim + MPC)
step c4_bench 240 bash -c "python3 bench.py --config 4 --cpu-budget 8 > $OUT/c4/bench.json 2> $OUT/c4/bench.err"
step c4_kt 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4/kt -- python3 bench.py --config 4 --no-per-env --no-h20 --no-cpu-baseline --no-single-step --no-other-configs > $OUT/c4/kt.log 2>&1
step c4_pmc 120 rocprofv3 --pmc $SQ --output-format csv -d $OUT/c4/pmc_sq -- python3 bench.py --config 4 --no-per-env --no-h20 --no-cpu-baseline --no-single-step --no-other-configs --steps 5 --warmup 2 --steady-launches 0 --settle-launches 0 > $OUT/c4/pmc_sq.log 2>&1
# dynamics-model training steps and the data-collection kernels
step train_kt 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train/kt -- python3 tools/exp_dyn_train.py > $OUT/train/out.txt 2>&1
step dataset_kt 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dataset/kt -- python3 tools/exp_dataset.py > $OUT/dataset/out.txt 2>&1
cut -c1-400 $OUT/c2/bench.json
