#!/bin/bash
# round-3 diagnostics in one GPU call: simulation-kernel stamps (H = 4 and 20), shared-reciprocal tanh A/B (config 3),
# the wide learner's grid, the config-3 issue-cycle PMC pass
set -u
export TMPDIR=/tmp
O=gpurun_out/r03_diag; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step clk4 120 bash -c "python3 tools/exp_dyn_clock.py tools/_build/libssc_clk.so 4 > $O/dyn_clock_h4.txt 2>&1; cat $O/dyn_clock_h4.txt"
step clk20 120 bash -c "python3 tools/exp_dyn_clock.py tools/_build/libssc_clk.so 20 > $O/dyn_clock_h20.txt 2>&1; cat $O/dyn_clock_h20.txt"
step wide 300 bash -c "python3 tools/exp_ddpg_wide.py > $O/ddpg_wide.txt 2>&1; cat $O/ddpg_wide.txt"
step ab 900 bash tools/gpu_variant_ab.sh 3 tools/_build/libssc_act_sharedrcp.so
step pmc 400 bash tools/profile_pmc.sh 3 $O/pmc_c3
