#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c26; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
for sh in prose prose; do timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | cut -c1-600 | tee -a $out/stage_times.txt; done
timeout -k 10 380 python3 tools/fuzz_hunt.py 5000 5100 2>&1 | grep -v "^seed .* done" | tail -3
