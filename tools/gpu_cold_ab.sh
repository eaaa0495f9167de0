#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/cold_ab; mkdir -p $O
for rep in 1 2 3; do
  for V in base tools/_build/libssc_bufall.so tools/_build/libssc_defer.so; do
    n=$(basename $V .so)
    if [ $V = base ]; then CMD="python3 bench.py"; else CMD="python3 tools/bench_with_lib.py $V"; fi
    timeout -k 10 120 $CMD --no-cpu-baseline --no-single-step --no-other-configs --steady-launches 0 > $O/$n$rep.json 2>/dev/null || { echo "step failed/killed"; exit 1; }
    python3 -c "import json;d=json.loads(open('$O/$n$rep.json').read().strip().splitlines()[-1]);print('$n', 'cold %.4f ms' % d['after_warmup_only']['ms_per_step'], 'settled %.4f ms' % d['ms_per_step'])"
    sleep 2
  done
done
