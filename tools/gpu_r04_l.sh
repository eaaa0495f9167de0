#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
step() { echo "== $1"; shift; timeout -k 10 "$@" || { echo "STEP FAILED ($?)"; exit 1; }; }
step tests 600 python -m pytest tests/test_gpu_navigator.py tests/test_gpu_smartstart_vec.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null
tail -2 $O/tests.log
export SSC_SMALL_SIM_SW=1
step tests_sw 600 python -m pytest tests/test_gpu_navigator.py tests/test_gpu_smartstart_vec.py -m gpu -x -q > $O/tests_sw.log 2>&1 < /dev/null
tail -2 $O/tests_sw.log
for sw in 0 1; do
export SSC_SMALL_SIM_SW=$sw
step time$sw 200 python3 tools/exp_small_sim.py 200 > $O/time$sw.txt 2>&1 < /dev/null
tail -1 $O/time$sw.txt
step time${sw}_n16 200 python3 tools/exp_small_sim.py 200 16 > $O/time${sw}_n16.txt 2>&1 < /dev/null
tail -1 $O/time${sw}_n16.txt
step pmc1 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc1 -- python3 tools/exp_small_sim.py 10 > $O/pmc1.log 2>&1 < /dev/null
python3 tools/pmc_by_kernel.py $O/pmc1 small_sim | tee $O/pmc1_sw$sw.txt
step pmc2 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -- python3 tools/exp_small_sim.py 10 > $O/pmc2.log 2>&1 < /dev/null
python3 tools/pmc_by_kernel.py $O/pmc2 small_sim | tee $O/pmc2_sw$sw.txt
rm -rf $O/pmc1 $O/pmc2
done
