#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c18; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py -m gpu -x -q -k "bucket_mode or streaming_machinery" > $out/tests_a.log 2>&1; rc=$?
tail -2 $out/tests_a.log
[ $rc = 0 ] || exit 1
for rep in 1 2 3; do
timeout -k 10 120 python3 tools/stage_times.py 256 random 8 2>/dev/null | tail -1 | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('random', d['ms_total'], d['ms_hist'], d['ms_pass_text'], d['ms_pass_rec'], d['ms_local_sort'])"
done
