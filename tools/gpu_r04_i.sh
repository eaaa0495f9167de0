#!/bin/bash
# round 4, call i: critic_l2_reg / clip_norm on the multi-workgroup learner, Dyn_Model.run_validation
mkdir -p gpurun_out/r04i
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_agents.py tests/test_gpu_navigator.py tests/test_gpu_layer_norm.py -m gpu -x -q -k "l2_reg or run_validation or ddpg_train or layer_norm" > gpurun_out/r04i/tests.log 2>&1 < /dev/null
tail -5 gpurun_out/r04i/tests.log
