#!/bin/bash
# in-kernel stamps of the simulation kernel (diagnostic build) at H = 4 and H = 20
set -u
export TMPDIR=/tmp
O=gpurun_out/c4_clk; mkdir -p $O
L=${1:-tools/_build/libssc_clk.so}
timeout -k 10 120 python3 tools/exp_dyn_clock.py $L 4 > $O/h4.txt 2>&1 || exit 1
timeout -k 10 120 python3 tools/exp_dyn_clock.py $L 20 > $O/h20.txt 2>&1 || exit 1
grep -v amdgpu.ids $O/h4.txt $O/h20.txt
