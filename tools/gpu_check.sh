#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/check
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 900 bash -c "python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
step smoke 200 bash -c "python3 -c 'import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")' > $O/smoke.log 2>&1; tail -2 $O/smoke.log"
step bench 300 bash -c "python3 bench.py > $O/bench.json 2> $O/bench.err; cut -c1-700 $O/bench.json"
