#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c20; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 tools/small_blocks.py 1 4 2>/dev/null | tee $out/small_blocks.txt | cut -c1-420
bash tools/timeline.sh random 4 && tail -45 gpurun_out/timeline_random.txt
