#!/bin/bash
# overlapped actor-learner loop: its test + the pipeline figures with and without overlap
set -u
export TMPDIR=/tmp
O=gpurun_out/overlap
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 300 bash -c "python3 -m pytest tests/test_gpu_agents.py -x -q -m gpu -k 'vec_ddpg' > $O/pytest.log 2>&1; tail -15 $O/pytest.log"
step pipe 300 bash -c "python3 tools/exp_pipeline.py > $O/pipeline.txt 2>$O/pipeline.err; cat $O/pipeline.txt; tail -3 $O/pipeline.err"
