#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c14; mkdir -p $out
bash tools/timeline.sh prose && bash tools/timeline.sh real || exit 1
for w in 2 3 4; do
  ARCHON_INV_WALK_WGS=$w timeout -k 10 200 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/wgs=$w /" | tee -a $out/inv.txt
done
for mb in 32 64 128; do for r in 0 1; do
  ARCHON_INV_ROWS=$r timeout -k 10 200 python3 tools/stage_times.py $mb random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/rows=$r mb=$mb /" | tee -a $out/inv.txt
done; done
