#!/bin/bash
# A/B of a variant library against the committed one on the same box: tools/gpu_variant_ab.sh <variant.so> [bench.py args]
# (config-2 bench line base/variant/base/variant, then the rollout parity tests with the variant in place of libssc.so)
set -u
export TMPDIR=/tmp
V=$1; shift
O=gpurun_out/variant_ab
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
line() { python3 -c "import json,sys;d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$1', d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_dist']['median'], r.get('steady',{}).get('median'), r.get('steady',{}).get('min'))"; }
for rep in 1 2; do
  step base$rep 200 bash -c "python3 bench.py --no-cpu-baseline --no-single-step $* > $O/base$rep.json 2>/dev/null"; line base$rep
  step var$rep 200 bash -c "python3 tools/bench_with_lib.py $V --no-cpu-baseline --no-single-step $* > $O/var$rep.json 2>/dev/null"; line var$rep
done
cp $V smartstartcontinuous_amd/libssc.so
step pytest 500 bash -c "python3 -m pytest tests/test_gpu_env.py tests/test_gpu_actor_pendulum.py tests/test_gpu_dataset.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log"
