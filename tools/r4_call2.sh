#!/bin/bash
# round 4, call 2: the mid-group rounds -- parity first, then times
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c2; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "mid_groups or segmented or period_defects or long_repeats or pair_chains" > $out/tests_mid.log 2>&1; rc=$?; echo "mid tests rc=$rc" | tee -a $out/rc.txt
tail -5 $out/tests_mid.log
[ $rc = 0 ] || exit 1
for sh in prose text motif_defects; do
  timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>$out/st_$sh.err | tail -1 | sed "s/^/$sh /" | tee -a $out/stage_times.txt
done
timeout -k 10 300 python3 tools/real_text.py 256 2>/dev/null | tail -1 | tee $out/real_text.json
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/rc.txt
tail -3 $out/tests.log
