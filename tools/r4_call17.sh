#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c17; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_inverse.py -x -q > $out/tests_inv.log 2>&1; rc=$?; echo "inv tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests_inv.log
[ $rc = 0 ] || exit 1
for r in 1 2 1 2; do
  ARCHON_INV_ROWS=$r timeout -k 10 200 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/rows=$r /" | tee -a $out/inv.txt
done
ARCHON_INV_ROWS=2 ARCHON_INV_WALK_WGS=2 timeout -k 10 200 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/rows=2 wgs=2 /" | tee -a $out/inv.txt
