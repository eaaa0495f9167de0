#!/bin/bash
# variant libraries: for a in 1 2 3: hipcc $BASEFLAGS -DSSC_DYN_ABLATE=$a -c csrc/dyn_mfma.hip -o _obj/dyn_mfma_abl$a.o; link with the other objects into tools/_build/libssc_dynabl$a.so
set -u
export TMPDIR=/tmp
O=gpurun_out/c4_abl; mkdir -p $O
for rep in 1 2; do
  for V in base tools/_build/libssc_dynabl1.so tools/_build/libssc_dynabl2.so tools/_build/libssc_dynabl3.so; do
    n=$(basename $V .so)
    if [ $V = base ]; then CMD="python3 bench.py"; else CMD="python3 tools/bench_with_lib.py $V"; fi
    timeout -k 10 150 $CMD --config 4 --no-cpu-baseline > $O/$n$rep.json 2>$O/$n$rep.err || { echo "$n failed"; tail -3 $O/$n$rep.err; continue; }
    python3 -c "import json;d=json.loads(open('$O/$n$rep.json').read().strip().splitlines()[-1]);r=d['roofline'];print('$n', 'mpc step %.4f ms' % d['ms_per_step'], 'sim %.4f ms' % r['kernel_ms'], 'b2b', r.get('kernel_ms_back_to_back'))"
  done
done
