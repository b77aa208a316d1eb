#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c6; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_line_$i.json 2> $out/bench.err; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c6/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['pipeline']['device_ms_per_block'], d['pipeline']['ms_hist'], [k['launch_ms'] for k in d['roofline']['kernels']])
PY
timeout -k 10 300 python3 tools/inv_sbits_sweep.py 2>$out/sweep.err | tee $out/inv_sbits_sweep.txt
timeout -k 10 200 python3 tools/stage_times.py 256 motif_defects 3 2>/dev/null | tail -1 | sed "s/^/motif_defects /" | tee -a $out/stage_times.txt
