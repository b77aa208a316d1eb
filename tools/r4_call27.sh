#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c27; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
for sh in prose dna prose; do timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | cut -c1-300 | tee -a $out/stage_times.txt; done
ARCHON_FORCE_PATH=0 timeout -k 10 200 python3 tools/stage_times.py 256 dna 3 2>/dev/null | tail -1 | cut -c1-300
