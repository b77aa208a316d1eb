#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c18; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
for sb in 8 7 8 7; do
  ARCHON_INV_SBITS=$sb timeout -k 10 200 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/sbits=$sb /" | tee -a $out/inv.txt
done
for mb in 128 64 16 4; do
  timeout -k 10 200 python3 tools/stage_times.py $mb random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/mb=$mb /" | tee -a $out/inv.txt
done
timeout -k 10 300 python3 tools/small_blocks.py 4 16 2>/dev/null | tee $out/small_blocks.txt | cut -c1-300
