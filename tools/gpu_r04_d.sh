#!/bin/bash
# round 4, call D: which vectorised-DDPG settings learn reliably (seeds 1..5 each)
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_d; mkdir -p $O
run() { timeout -k 10 120 python3 tools/exp_vec_learning.py ddpg "$@" 2>&1 | grep "^cfg" | tee -a $O/sweep.txt; }
for seed in 1 2 3 4 5; do
  run 4096 32 50 64 1000 $seed 0 20 50 1
  run 4096 32 50 64 1000 $seed 4 20 50 1
  run 4096 32 100 64 1000 $seed 4 20 50 1
  run 4096 16 50 64 2000 $seed 2 20 50 1
  run 4096 32 25 1024 1000 $seed 4 20 50 1
done
python3 tools/exp_c4_walk.py - 2>/dev/null | tee -a $O/ab.txt
python3 tools/exp_c4_walk.py tools/_build/libssc_c4old.so 2>/dev/null | tee -a $O/ab.txt
python3 tools/exp_c4_walk.py - 2>/dev/null | tee -a $O/ab.txt
