#!/bin/bash
O=gpurun_out/r04z; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 900 python -m pytest tests/test_gpu_smartstart_vec.py tests/test_gpu_smartstart_curves.py tests/test_gpu_vec_learning.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
for v in branch serial branch serial; do
  step ss_$v 300 python tools/prof_smartstart_vec.py 40 $v > $O/ss_$v.txt 2>&1 < /dev/null
  echo $v $(tail -1 $O/ss_$v.txt | grep -o '"ms_per_step": [0-9.]*')
done
step ex 300 bash -c "python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 20 --samples 16 --plans 8 2>&1 | grep -v 'amdgpu.ids\|RuntimeWarning\|self.pool.publish' | tail -2 > $O/ex.txt"
cut -c1-230 $O/ex.txt
