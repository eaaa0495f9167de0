#!/bin/bash
# PMC passes for the DDPG learner kernels (tools/exp_train.py) and the KDE kernel; separate passes, no tracing domains.
set -u
OUT=${1:-gpurun_out/pmc_learner}
export TMPDIR=/tmp
mkdir -p $OUT
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step a 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_a -- python3 tools/exp_train.py > $OUT/pmc_a.log 2>&1
step b 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_b -- python3 tools/exp_train.py > $OUT/pmc_b.log 2>&1
python3 tools/pmc_summary.py $OUT ddpg_train_fixed > $OUT/summary.txt 2>&1
python3 tools/pmc_summary.py $OUT kde_kernel >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
