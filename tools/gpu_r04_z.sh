#!/bin/bash
O=gpurun_out/r04z; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_navigator.py tests/test_gpu_dataset.py tests/test_gpu_smartstart_vec.py -m gpu -x -q -k "train or dynamics or aggregation or dataset or learn" > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
step train 300 python3 tools/exp_dyn_train.py > $O/train.txt 2>&1 < /dev/null
tail -12 $O/train.txt
