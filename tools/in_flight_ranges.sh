#!/bin/bash
# two blocks in flight: does cutting the passes into FEWER ranges than CUs (the two blocks' passes side by side on disjoint CUs) beat
# one range per CU?   usage: tools/in_flight_ranges.sh <out file>
out=${1:-gpurun_out/in_flight_ranges.txt}
for r in 256 128 160 192 224 256; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --pass-ranges $r --no-shapes --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('pass_ranges', d['config']['pass_ranges'], 'in flight', d['config']['blocks_in_flight'], d['value'], 'MB/s', d['ms_per_step'], 'ms | one at a time', d['one_block_at_a_time']['ms_per_step'], '| pass A', d['pipeline']['ms_pass_text'], 'pass B', d['pipeline']['ms_pass_rec'])" >> $out || exit 1
done
cat $out
