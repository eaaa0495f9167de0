#!/bin/bash
# round-2 first GPU call: sanity tests, drift study, the new bench line, stall-split counters of configs 2 and 3.
# A step that is killed at its limit ends the call (no further GPU step after a timeout).
set -u
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
step() { # name, limit, command...
  local name=$1 lim=$2; shift 2
  timeout -k 10 $lim "$@"; local rc=$?
  echo "[$name] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi
}
step pytest 600 bash -c "python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log"
step drift 300 bash -c "python3 tools/exp_drift.py --out $O/drift.json > $O/drift.log 2>&1; tail -20 $O/drift.log"
step bench2 300 bash -c "python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-1800 $O/bench_c2.json"
step bench3 200 bash -c "python3 bench.py --config 3 --cpu-budget 6 > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-600 $O/bench_c3.json"
step pmc3 400 bash -c "tools/profile_pmc.sh 3 $O/pmc_c3 > $O/pmc_c3.log 2>&1; tail -30 $O/pmc_c3.log"
step pmc2 400 bash -c "tools/profile_pmc.sh 2 $O/pmc_c2 > $O/pmc_c2.log 2>&1; tail -30 $O/pmc_c2.log"
