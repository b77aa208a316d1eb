#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c17; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py -m gpu -x -q -k "compact or alphabet or dna or shapes_vs_oracle or bucket_mode or natural or both_paths" > $out/tests_a.log 2>&1; rc=$?
tail -2 $out/tests_a.log
[ $rc = 0 ] || exit 1
for sh in dna random; do
timeout -k 10 120 python3 tools/stage_times.py 256 $sh 5 2>/dev/null | tail -1 | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$sh', d['ms_total'], d['ms_hist'], d['ms_pass_text'], d['ms_pass_rec'], d['ms_local_sort'])"
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_golden.py -m gpu -x -q -k "dna" > $out/tests_g.log 2>&1; tail -2 $out/tests_g.log
