#!/bin/bash
O=gpurun_out/r04r; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_agents.py tests/test_gpu_layer_norm.py tests/test_gpu_navigator.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step iter 300 python tools/exp_learner_iter.py > $O/learner_iter.txt 2>&1 < /dev/null
grep batch $O/learner_iter.txt
step loop 600 python tools/prof_vec_ddpg_loop.py > $O/vec_ddpg_loop.txt 2>&1 < /dev/null
tail -1 $O/vec_ddpg_loop.txt
step multirank 600 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_vec_learning.py -m gpu -x -q > $O/tests2.log 2>&1 < /dev/null
tail -3 $O/tests2.log
