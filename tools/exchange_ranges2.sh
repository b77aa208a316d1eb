#!/bin/bash
# as tools/exchange_ranges.sh, with pass B by ranges (pass_b_buckets 0: what N > 1 runs) and by buckets
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-xr2}; mkdir -p $out
for cfg in "-1 1024 0" "-1 224 0" "-1 224 1" "-1 192 0" "-1 240 0" "0 224 0" "0 1024 0"; do
  set -- $cfg
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 24 --warmup 5 --no-cpu-baseline --no-shapes --gather-batch $1 --pass-ranges $2 --pass-b-buckets $3 2> $out/err.txt | tail -1 > $out/line.json || { tail -20 $out/err.txt; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/line.json'))
print('gather_batch $1 pass_ranges $2 pass_b_buckets $3: %.1f MB/s %.3f ms per step | one at a time %s | gates %s' % (d['value'], d['ms_per_step'], d.get('one_block_at_a_time',{}).get('ms_per_step'), d['config']['gates_passed']))" | tee -a $out/exchange_ranges.txt
done
timeout -k 10 500 python3 tools/key_bytes_sweep.py prose text real > $out/key_bytes_sweep.txt 2> $out/kb.err || { tail -5 $out/kb.err; }
cat $out/key_bytes_sweep.txt
