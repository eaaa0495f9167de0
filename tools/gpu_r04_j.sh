#!/bin/bash
# round 4, call j: the shipped networks on batches of 64-row tiles
mkdir -p gpurun_out/r04j
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_agents.py tests/test_gpu_layer_norm.py -m gpu -x -q > gpurun_out/r04j/tests.log 2>&1 < /dev/null
tail -3 gpurun_out/r04j/tests.log
step iter 300 python tools/exp_learner_iter.py > gpurun_out/r04j/learner_iter.txt 2>&1 < /dev/null
cat gpurun_out/r04j/learner_iter.txt
step loop 600 python tools/prof_vec_ddpg_loop.py > gpurun_out/r04j/vec_ddpg_loop.txt 2>&1 < /dev/null
cat gpurun_out/r04j/vec_ddpg_loop.txt
