#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of bench.py.
# usage: tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --no-cpu-baseline --no-other-configs > $OUT/kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 > $OUT/pmc_fetch.log 2>&1
cat $OUT/bench.json
