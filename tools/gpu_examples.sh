#!/bin/bash
# the three examples end to end (bit-rot check)
set -u
export TMPDIR=/tmp
O=gpurun_out/examples; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "[$name] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit 1; fi; }
step ddpg_vec 200 bash -c "python3 examples/ddpg_mountaincar.py --mode vec --chunks 6 > $O/ddpg_vec.txt 2>&1; tail -3 $O/ddpg_vec.txt"
step ddpg_single 200 bash -c "python3 examples/ddpg_mountaincar.py --mode single --episodes 1 > $O/ddpg_single.txt 2>&1; tail -3 $O/ddpg_single.txt"
step smartstart 300 bash -c "python3 examples/smartstart_ddpg.py > $O/smartstart.txt 2>&1; tail -4 $O/smartstart.txt"
step navigator 300 bash -c "python3 examples/navigator_from_scratch.py > $O/navigator.txt 2>&1; tail -4 $O/navigator.txt"
