#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step pmc1 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- python3 bench.py --config 5 --steps 5 --warmup 2 --settle-launches 0 > $O/pmc1.log 2>&1 < /dev/null
python3 tools/pmc_by_kernel.py $O/pmc1 mpc_pass | tee $O/pmc1.txt
step pmc2 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $O/pmc2 -- python3 bench.py --config 5 --steps 5 --warmup 2 --settle-launches 0 > $O/pmc2.log 2>&1 < /dev/null
python3 tools/pmc_by_kernel.py $O/pmc2 mpc_pass | tee $O/pmc2.txt
step pmc3 300 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAVES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc3 -- python3 bench.py --config 5 --steps 5 --warmup 2 --settle-launches 0 > $O/pmc3.log 2>&1 < /dev/null
python3 tools/pmc_by_kernel.py $O/pmc3 mpc_pass | tee $O/pmc3.txt
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
