#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c9; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/c9/bench_line.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline']['device_ms_per_block'], d['pipeline']['arena_bytes_per_input_byte'])
PY
for sb in 6 7 8; do ARCHON_INV_SBITS=$sb timeout -k 10 200 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | tail -1 | sed "s/^/inverse-256MiB sbits=$sb /" | tee -a $out/inv_sbits_256.txt; done
for sh in prose text; do timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | sed "s/^/$sh /" | tee -a $out/stage_times.txt; done
bash tools/ab_stage.sh "$PWD/dark-archon_amd/libarchon_hip.so $PWD/dark-archon_amd/libarchon_hip_fu512.so" prose 256 2 2>/dev/null | python3 -c "
import sys, json
lib=None
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): lib=l.split('/')[-1]
    elif l.startswith('{'): d=json.loads(l); print('prose', lib, d['ms_total'], d.get('mid_items'))
" | tee -a $out/ab_fu512.txt
