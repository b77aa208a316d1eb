#!/bin/bash
# one rank under torch.distributed.run over RCCL: the batched exchange forced (a 268 MB self-exchange per step beside the sorts) against
# the number of pass ranges -- what RCCL's resident kernels cost a pass that wants every CU
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-xr}; mkdir -p $out
for cfg in "-1 256" "-1 1024" "-1 224" "0 1024" "0 256"; do
  set -- $cfg
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 24 --warmup 5 --no-cpu-baseline --no-shapes --gather-batch $1 --pass-ranges $2 2> $out/err.txt | tail -1 > $out/line.json || { tail -20 $out/err.txt; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/line.json'))
print('gather_batch $1 pass_ranges $2: %.1f MB/s %.3f ms per step | one at a time %s | gates %s' % (d['value'], d['ms_per_step'], d.get('one_block_at_a_time',{}).get('ms_per_step'), d['config']['gates_passed']))" | tee -a $out/exchange_ranges.txt
done
