#!/bin/bash
# parity tests of the navigator / dynamics kernels with the committed library, then the config-4 A/B against variant libraries
set -u
export TMPDIR=/tmp
O=gpurun_out/c4_lag; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -ne 0 ]; then echo "[$name] failed: stopping"; exit 1; fi; }
step pytest 400 bash -c "python3 -m pytest tests/test_gpu_navigator.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; rc=\$?; tail -15 $O/pytest.log; exit \$rc"
step ab 900 bash tools/gpu_c4_ab.sh "$@"
