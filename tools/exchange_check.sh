#!/bin/bash
# the batched exchange of bench.py's N > 1 path on a one-GPU box: bench.py with 2 and 3 ranks over gloo (tests), one rank under
# torch.distributed.run over RCCL with per-step gathers (the default at one rank) and with the batched exchange forced (--gather-batch -1)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-exchange}; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_shard.py -x -q -m gpu -k "rehearsal or default_run or shard" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for b in 0 -1; do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-shapes --gather-batch $b 2> $out/w1_batch$b.err | tail -1 > $out/bench_line_torchrun_world1_batch$b.json || { tail -20 $out/w1_batch$b.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$out/bench_line_torchrun_world1_batch$b.json'))
print('batch $b', d['value'], d['ms_per_step'], d['config']['gates_passed'], d['config']['gathered_block_round_trip'], d['config']['exchange'][:60])"
done
