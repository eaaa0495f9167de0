#!/bin/bash
# round 4, call A: the walking simulation kernel (parity, config-4 bench with the per-env leg, in-kernel stamps at H = 4 / 20 and
# for the walking launch) and the host-free actor-learner loop
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_a; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step tests 600 bash -c "python3 -m pytest tests/test_gpu_navigator.py tests/test_gpu_smartstart_vec.py tests/test_gpu_navigator_runs.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log"
step c4 300 bash -c "python3 bench.py --config 4 --no-cpu-baseline > $O/c4.json 2> $O/c4.err; python3 - <<'PY'
import json
for l in open('$O/c4.json'):
    if l.startswith('{'):
        r = json.loads(l)
        print(r['metric'][:60], 'ms/step %.4f' % r['ms_per_step'], 'frac %.3f' % r['roofline']['frac'], 'kernel_ms', r['roofline'].get('kernel_ms'), 'h20', r.get('h20', {}).get('frac'))
PY"
step clk4 120 bash -c "python3 tools/exp_dyn_clock.py tools/_build/libssc_clk.so 4 sample 2>&1 | grep -v amdgpu.ids > $O/clk_h4.txt; cat $O/clk_h4.txt"
step clk20 120 bash -c "python3 tools/exp_dyn_clock.py tools/_build/libssc_clk.so 20 sample 2>&1 | grep -v amdgpu.ids > $O/clk_h20.txt; cat $O/clk_h20.txt"
step clkwalk 120 bash -c "python3 tools/exp_dyn_clock.py tools/_build/libssc_clk.so 4 sample 1048576 2>&1 | grep -v amdgpu.ids > $O/clk_walk.txt; cat $O/clk_walk.txt"
step loop 300 bash -c "python3 tools/exp_pipeline.py wide 2>&1 | grep -v amdgpu.ids > $O/vec_ddpg_loop.txt; cat $O/vec_ddpg_loop.txt"
