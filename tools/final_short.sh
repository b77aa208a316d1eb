#!/bin/bash
# the short form of tools/final_suite.sh for a call of a few minutes: the driver's own bench command, then the whole -m gpu suite
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-final_short}; mkdir -p $out
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$out/bench_line.json'))
print('bench', d['value'], d['ms_per_step'], d['config']['gates_passed'], d['one_block_at_a_time'], d['inverse']['ms'] if 'ms' in d['inverse'] else d['inverse'])"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $out/tests_final.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -2 $out/tests_final.log
exit $rc
