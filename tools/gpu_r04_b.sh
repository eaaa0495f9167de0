#!/bin/bash
# round 4, call B: walking simulation kernel -- parity, then A/B against the round-3 kernel (tools/_build/libssc_c4old.so), interleaved
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_b; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step tests 600 bash -c "python3 -m pytest tests/test_gpu_navigator.py tests/test_gpu_smartstart_vec.py -m gpu -q > $O/tests.log 2>&1; tail -6 $O/tests.log"
for rep in 1 2 3; do
  step new$rep 120 bash -c "python3 tools/exp_c4_walk.py - 2>/dev/null | tee -a $O/ab.txt"
  step old$rep 120 bash -c "python3 tools/exp_c4_walk.py tools/_build/libssc_c4old.so 2>/dev/null | tee -a $O/ab.txt"
done
step clkwalk 120 bash -c "python3 tools/exp_dyn_clock.py tools/_build/libssc_clk.so 4 sample 1048576 2>&1 | grep -v amdgpu.ids > $O/clk_walk.txt; cat $O/clk_walk.txt"
step c4 300 bash -c "python3 bench.py --config 4 --no-cpu-baseline > $O/c4.json 2> $O/c4.err; tail -c 600 $O/c4.json"
