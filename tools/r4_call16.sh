#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c16; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_inverse.py -x -q > $out/tests_inv.log 2>&1; rc=$?; echo "inv tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests_inv.log
[ $rc = 0 ] || exit 1
for mb in 256 64 16 4; do
  timeout -k 10 200 python3 tools/stage_times.py $mb random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/mb=$mb /" | tee -a $out/inv.txt
done
ARCHON_INV_SBITS=7 timeout -k 10 200 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/sbits7 /" | tee -a $out/inv.txt
timeout -k 10 300 python3 tools/inv_exp.py 2>/dev/null | tee $out/inv_exp.txt
bash tools/inv_kernels.sh 8 7 | tee $out/inv_kernels.txt
