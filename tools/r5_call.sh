#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c14; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py -m gpu -x -q -k "bucket_mode or streaming_machinery or natural_route or container or post" > $out/tests_a.log 2>&1; rc=$?
tail -2 $out/tests_a.log
[ $rc = 0 ] || exit 1
for mib in 16 64 128 256; do
  timeout -k 10 120 python3 tools/stage_times.py $mib random 4 2>/dev/null | tail -1 | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$mib MiB', d['ms_total'], d['ms_hist'], d['ms_pass_text'], d['ms_pass_rec'], d['ms_local_sort'], d['kernel_launches'])"
done
timeout -k 10 120 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | tail -1 | cut -c1-200
