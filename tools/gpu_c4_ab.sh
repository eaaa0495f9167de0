#!/bin/bash
# config-4 bench line (MPC step, simulation kernel) for the committed library and variant builds, interleaved:
#   tools/gpu_c4_ab.sh <variant.so> [...]; then the navigator parity tests with the LAST variant loaded through SSC_LIB_PATH (libssc.so itself is never overwritten)
set -u
export TMPDIR=/tmp
O=gpurun_out/c4_ab; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
line() { python3 -c "import json;d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$1', 'mpc step %.4f ms' % d['ms_per_step'], 'sim %.4f ms' % r['kernel_ms'], 'b2b', r.get('kernel_ms_back_to_back'))"; }
for rep in 1 2 3; do
  step base$rep 200 bash -c "python3 bench.py --config 4 --no-per-env --no-cpu-baseline > $O/base$rep.json 2>/dev/null"; line base$rep
  for V in "$@"; do
    n=$(basename $V .so)
    step $n$rep 200 bash -c "python3 tools/bench_with_lib.py $V --config 4 --no-per-env --no-cpu-baseline > $O/$n$rep.json 2>/dev/null"; line $n$rep
  done
done
for V in "$@"; do :; done
step pytest 500 bash -c "SSC_LIB_PATH=$PWD/$V python3 -m pytest tests/test_gpu_navigator.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
