#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c19; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 tools/small_blocks.py 1 4 16 2>/dev/null | tee $out/small_blocks.txt | cut -c1-420
for sh in prose text; do timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sh', d['ms_total'])" | tee -a $out/stage_times.txt; done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
