#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c13; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_inverse.py -x -q > $out/tests_inv.log 2>&1; rc=$?; echo "inv tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests_inv.log
[ $rc = 0 ] || exit 1
for r in 1 0 1 0; do
  for mb in 256 16 4; do
    ARCHON_INV_ROWS=$r timeout -k 10 200 python3 tools/stage_times.py $mb random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-300 | sed "s/^/rows=$r mb=$mb /" | tee -a $out/inv.txt
  done
done
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
