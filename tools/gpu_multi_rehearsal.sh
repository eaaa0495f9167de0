#!/bin/bash
# N > 1 control flow of bench.py on a one-GPU box: (1) a 1-rank RCCL group (--force-dist), (2) two gloo ranks sharing the
# GPU under torch.distributed.run, (3) the same started from a BARE `bench.py --gpus 2` (self-spawned launcher)
set -u
export TMPDIR=/tmp
O=gpurun_out/multi; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-single-step --no-other-configs"
step rccl1 300 bash -c "python3 bench.py --force-dist $COMMON > $O/bench_rccl_1rank.json 2> $O/rccl1.err; tail -c 900 $O/bench_rccl_1rank.json | cut -c1-400"
step gloo2 400 bash -c "python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --envs-per-gpu 16384 $COMMON > $O/bench_gloo_2ranks_1gpu.json 2> $O/gloo2.err; tail -c 900 $O/bench_gloo_2ranks_1gpu.json | cut -c1-400"
step gloo2_bare 400 bash -c "python3 bench.py --gpus 2 --backend gloo --envs-per-gpu 16384 $COMMON > $O/bench_gloo_2ranks_selfspawn.json 2> $O/gloo2b.err; tail -c 900 $O/bench_gloo_2ranks_selfspawn.json | cut -c1-400"
