#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step kt 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 examples/smartstart_ddpg.py --mode vec --max-steps 300 --envs 65536 --chunks 12 --samples 16 --plans 8 > $O/out.txt 2>&1 < /dev/null
f=$(ls $O/kt/*/*_kernel_stats.csv | head -1)
[ -n "$f" ] && cut -c1-150 "$f" | head -24 | tee $O/kernel_stats_head.txt
cp "$f" $O/kernel_stats.csv
rm -rf $O/kt
