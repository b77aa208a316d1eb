#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c5; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_golden.py -m gpu -x -q -k "small_blocks_product or reference_small or mid_groups or period_defects" > $out/tests_quick.log 2>&1; rc=$?; echo "quick tests rc=$rc" | tee -a $out/rc.txt
tail -5 $out/tests_quick.log
[ $rc = 0 ] || exit 1
timeout -k 10 400 python3 tools/small_blocks.py 1 4 7 2>$out/small.err | tee $out/small_blocks.txt
ARCHON_SMALL_BLOCK=0 timeout -k 10 400 python3 tools/small_blocks.py 4 2>>$out/small.err | sed 's/^/streaming-stage /' | tee -a $out/small_blocks.txt
for sh in motif_defects prose; do
  timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>$out/st_$sh.err | tail -1 | sed "s/^/$sh /" | tee -a $out/stage_times.txt
done
timeout -k 10 300 python3 tools/two_ctx.py 256 6 > $out/two_contexts.txt 2>/dev/null; cat $out/two_contexts.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/rc.txt
tail -3 $out/tests.log
