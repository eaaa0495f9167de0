#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/learner_check
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 300 bash -c "python3 -m pytest tests/test_gpu_agents.py -x -q -m gpu -k 'ddpg or DDPG or sharded or pipelined' > $O/pytest.log 2>&1; tail -30 $O/pytest.log"
step phases 120 bash -c "python3 tools/exp_ddpg_phases.py > $O/ddpg_phases_fixed.txt 2>&1; cat $O/ddpg_phases_fixed.txt"
step train 120 bash -c "python3 tools/exp_train.py > $O/train.txt 2>&1; tail -4 $O/train.txt"
