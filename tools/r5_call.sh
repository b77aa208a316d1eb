#!/bin/bash
# scratch driver of one gpurun call (rewritten per call; the kept evidence recipe is tools/evidence.sh)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c1; mkdir -p $out
timeout -k 10 500 python3 -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "clean_periodic or closed_form or long_repeats or periodic_blocks or known_answers" > $out/tests_a.log 2>&1; rc=$?
tail -5 $out/tests_a.log
[ $rc = 0 ] || exit 1
for sh in a ab motif random; do
  timeout -k 10 120 python3 tools/stage_times.py 256 $sh 4 2>$out/err_$sh.txt | tail -1 | sed "s/^/$sh /" | tee -a $out/stage_times.txt | cut -c1-400 || exit 1
done
timeout -k 10 500 python3 -m pytest tests/test_gpu_golden.py -m gpu -x -q > $out/tests_golden.log 2>&1; rc=$?
tail -3 $out/tests_golden.log
[ $rc = 0 ] || exit 1
