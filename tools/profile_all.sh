#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats (+ PMC where it matters) for the
# three measured configurations.  usage: tools/profile_all.sh <tag>  -> gpurun_out/prof_<tag>/{c2,c3,c4}
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT/c2 $OUT/c3 $OUT/c4 $OUT/train $OUT/dataset
# config 2 (headline)
python3 bench.py > $OUT/c2/bench.json 2> $OUT/c2/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2/kt -- python3 bench.py --no-cpu-baseline > $OUT/c2/kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c2/pmc_write -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c2/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c2/pmc_fetch.log 2>&1
# config 3 (actor MFMA rollout)
python3 bench.py --config 3 --cpu-budget 8 > $OUT/c3/bench.json 2> $OUT/c3/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3/kt -- python3 bench.py --config 3 --no-cpu-baseline > $OUT/c3/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/c3/pmc_sq -- python3 bench.py --config 3 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c3/pmc_sq.log 2>&1
# config 4 (dynamics MLP forward sim + MPC)
python3 bench.py --config 4 --cpu-budget 8 > $OUT/c4/bench.json 2> $OUT/c4/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4/kt -- python3 bench.py --config 4 --no-cpu-baseline > $OUT/c4/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/c4/pmc_sq -- python3 bench.py --config 4 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/c4/pmc_sq.log 2>&1
# dynamics-model training steps and the data-collection kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train/kt -- python3 tools/exp_dyn_train.py > $OUT/train/out.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dataset/kt -- python3 tools/exp_dataset.py > $OUT/dataset/out.txt 2>&1
cat $OUT/c2/bench.json | cut -c1-300
