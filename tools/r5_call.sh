#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c12; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "bucket_mode" > $out/tests_a.log 2>&1; rc=$?
tail -2 $out/tests_a.log
[ $rc = 0 ] || exit 1
for rep in 1 2 3; do for v in 0 1; do
ARCHON_NO_REL_RECORDS=$v timeout -k 10 120 python3 tools/stage_times.py 256 random 8 2>/dev/null | tail -1 | sed "s/^/norel=$v /" | tee -a $out/stage_times.txt | python3 -c "
import sys,json
for l in sys.stdin:
    t,j=l.split(' ',1); d=json.loads(j); print(t, d['ms_total'], d['ms_hist'], d['ms_pass_text'], d['ms_pass_rec'], d['ms_local_sort'], d['ms_resolve'])"
done; done
timeout -k 10 200 python3 tools/pass_stamps.py 256 random 2>&1 | grep -A12 "pass B" | head -14
timeout -k 10 300 python3 -m pytest tests/test_gpu_golden.py -m gpu -x -q -k "full_size" > $out/tests_g.log 2>&1; tail -2 $out/tests_g.log
