#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c13; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_fuzz.py -m gpu -x -q > $out/tests_a.log 2>&1; rc=$?
tail -2 $out/tests_a.log
[ $rc = 0 ] || exit 1
for sh in prose random_copy text; do for v in 0 1; do
ARCHON_NO_LADDER=$v timeout -k 10 200 python3 tools/stage_times.py 256 $sh 4 2>/dev/null | tail -1 | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$sh noladder=$v', d['ms_total'], 'rounds', d['doubling_rounds'], 'ladder', d.get('ladder_rounds'), 'syncs', d.get('host_syncs'), 'launches', d['kernel_launches'])" | tee -a $out/ladder.txt
done; done
timeout -k 10 300 python3 tools/real_text.py 256 2>/dev/null | tail -1 | cut -c1-300 | tee -a $out/ladder.txt
timeout -k 10 400 python3 -m pytest tests/test_gpu_golden.py -m gpu -x -q > $out/tests_g.log 2>&1; tail -2 $out/tests_g.log
