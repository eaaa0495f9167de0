#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 240 bash -c "python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log"
step bench2 300 bash -c "python3 bench.py --cpu-budget 8 > $O/bench_c2.json 2> $O/bench_c2.err; python3 -c \"import json;d=json.load(open('$O/bench_c2.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel_ms_dist'],d['roofline']['steady'])\""
step bench3 200 bash -c "python3 bench.py --config 3 --cpu-budget 4 --steps 100 > $O/bench_c3.json 2> $O/bench_c3.err; python3 -c \"import json;d=json.load(open('$O/bench_c3.json'));print(d['value'],d['roofline']['kernel_ms_dist'])\""
step pmc3 240 bash -c "tools/profile_pmc.sh 3 $O/pmc_c3 > $O/pmc_c3.log 2>&1; tail -26 $O/pmc_c3.log"
step align 300 bash -c "python3 tools/exp_align.py > $O/align.txt 2>&1; cat $O/align.txt"
