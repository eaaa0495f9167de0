#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r02g
mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step pytest 300 bash -c "python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log"
step gloo2 240 bash -c "python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --settle-launches 20 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -3 $O/bench_gloo2.err; cut -c1-900 $O/bench_gloo2.json"
step gloo2full 240 bash -c "python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --backend gloo --gather full --chunk 64 --steps 4 --warmup 2 --settle-launches 2 > $O/bench_gloo2_full.json 2> $O/bench_gloo2_full.err; tail -3 $O/bench_gloo2_full.err; cut -c1-700 $O/bench_gloo2_full.json"
step rccl1 240 bash -c "python3 bench.py --force-dist --no-cpu-baseline --no-single-step > $O/bench_rccl1.json 2> $O/bench_rccl1.err; tail -2 $O/bench_rccl1.err; cut -c1-900 $O/bench_rccl1.json"
