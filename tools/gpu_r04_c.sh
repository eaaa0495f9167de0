#!/bin/bash
# round 4, call C: WALK instantiation regression + do the vectorised loops learn (tools/exp_vec_learning.py)
set -u
export TMPDIR=/tmp
O=gpurun_out/r04_c; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step tests 600 bash -c "python3 -m pytest tests/test_gpu_navigator.py tests/test_gpu_agents.py -m gpu -q > $O/tests.log 2>&1; tail -4 $O/tests.log"
step new 120 bash -c "python3 tools/exp_c4_walk.py - 2>/dev/null | tee -a $O/ab.txt"
step old 120 bash -c "python3 tools/exp_c4_walk.py tools/_build/libssc_c4old.so 2>/dev/null | tee -a $O/ab.txt"
step new 120 bash -c "python3 tools/exp_c4_walk.py - 2>/dev/null | tee -a $O/ab.txt"
step ddpg1 200 bash -c "python3 tools/exp_vec_learning.py ddpg 4096 128 100 64 200 1 2>&1 | grep -v amdgpu.ids | tee $O/ddpg_4096_128_100x64_s1.txt"
step ddpg2 200 bash -c "python3 tools/exp_vec_learning.py ddpg 4096 128 25 1024 200 1 2>&1 | grep -v amdgpu.ids | tee $O/ddpg_4096_128_25x1024_s1.txt"
step ddpg3 200 bash -c "python3 tools/exp_vec_learning.py ddpg 4096 32 50 64 800 2 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/ddpg_4096_32_50x64_s2.txt"
step ss1 300 bash -c "python3 tools/exp_vec_learning.py ss 65536 16 20 1234 0 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/ss_1234_0.txt"
step ss2 300 bash -c "python3 tools/exp_vec_learning.py ss 65536 16 20 1234 1 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/ss_1234_1.txt"
step ss3 300 bash -c "python3 tools/exp_vec_learning.py ss 65536 16 20 7 0 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/ss_7_0.txt"
