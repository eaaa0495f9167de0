#!/bin/bash
# A/B of variant libraries against the committed one on the same box:
#   tools/gpu_variant_ab.sh <config: 2|3> <variant.so> [<variant2.so> ...]
# (bench line base / variants, interleaved twice; then the rollout parity tests with the LAST variant loaded through SSC_LIB_PATH (libssc.so itself is never overwritten))
set -u
export TMPDIR=/tmp
C=$1; shift
O=gpurun_out/variant_ab
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
line() { python3 -c "import json,sys;d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$1', '%.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'kernel %.4f' % r['kernel_ms'], 'median %.4f' % r['kernel_ms_dist']['median'], 'steady', r.get('steady',{}).get('median'), r.get('steady',{}).get('min'))"; }
ARGS="--config $C --no-cpu-baseline"
[ "$C" = 2 ] && ARGS="--no-cpu-baseline --no-single-step --no-other-configs"
for rep in 1 2; do
  step base$rep 200 bash -c "python3 bench.py $ARGS > $O/c${C}_base$rep.json 2>/dev/null"; line c${C}_base$rep
  for V in "$@"; do
    n=$(basename $V .so)
    step $n$rep 200 bash -c "python3 tools/bench_with_lib.py $V $ARGS > $O/c${C}_$n$rep.json 2>/dev/null"; line c${C}_$n$rep
  done
done
for V in "$@"; do :; done
step pytest 500 bash -c "SSC_LIB_PATH=$PWD/$V python3 -m pytest tests/test_gpu_env.py tests/test_gpu_actor_pendulum.py tests/test_gpu_dataset.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
