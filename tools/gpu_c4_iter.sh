#!/bin/bash
# one iteration of the config-4 kernel work: navigator parity tests, stamps (H = 4), A/B against variant libraries
set -u
export TMPDIR=/tmp
O=gpurun_out/c4_iter; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -ne 0 ]; then echo "[$name] failed: stopping"; exit 1; fi; }
step pytest 400 bash -c "python3 -m pytest tests/test_gpu_navigator.py -x -q -m gpu > $O/pytest.log 2>&1; rc=\$?; tail -5 $O/pytest.log; exit \$rc"
step clk 200 bash -c "python3 tools/exp_dyn_clock.py tools/_build/libssc_clk.so 4 sample 2>&1 | grep -v amdgpu.ids | tee $O/h4.txt"
line() { python3 -c "
import json
d=json.loads(open('$O/$1.json').read().strip().splitlines()[0]);r=d['roofline'];print('$1', 'mpc step %.4f ms' % d['ms_per_step'], 'sim %.4f ms' % r['kernel_ms'], 'b2b', r.get('kernel_ms_back_to_back'))"; }
for rep in 1 2 3; do
  step base$rep 200 bash -c "python3 bench.py --config 4 --no-per-env --no-cpu-baseline > $O/base$rep.json 2>/dev/null"; line base$rep
  for V in "$@"; do
    n=$(basename $V .so)
    step $n$rep 200 bash -c "python3 tools/bench_with_lib.py $V --config 4 --no-per-env --no-cpu-baseline > $O/$n$rep.json 2>/dev/null"; line $n$rep
  done
done
