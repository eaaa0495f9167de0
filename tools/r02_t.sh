#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02t
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 400 bash -c "python3 -m pytest tests/test_gpu_agents.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log"
step pipeline 300 bash -c "python3 tools/exp_pipeline.py > $O/pipeline.txt 2>&1; cat $O/pipeline.txt"
