#!/bin/bash
# Stall-split PMC passes for one bench.py config (run on the GPU box via gpurun; separate passes, no tracing
# domains beside --pmc).  usage: tools/profile_pmc.sh <config 2|3|4> <outdir>
set -u
CFG=${1:-3}
OUT=${2:-gpurun_out/pmc_c$CFG}
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--config $CFG --no-cpu-baseline --steps 5 --warmup 2 --steady-launches 0 --no-single-step --no-other-configs"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA \
  --output-format csv -d $OUT/pmc_a -- python3 bench.py $ARGS > $OUT/pmc_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
  --output-format csv -d $OUT/pmc_b -- python3 bench.py $ARGS > $OUT/pmc_b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM \
  --output-format csv -d $OUT/pmc_c -- python3 bench.py $ARGS > $OUT/pmc_c.log 2>&1
python3 tools/pmc_summary.py $OUT rollout > $OUT/summary.txt 2>&1
python3 tools/pmc_summary.py $OUT dyn_mfma >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
