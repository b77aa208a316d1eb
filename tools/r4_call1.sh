#!/bin/bash
# round 4, call 1: the whole GPU suite, the bench line, the one-rank RCCL line with and without channel limits, validate timing
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c1; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/rc.txt
tail -3 $out/tests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/rc.txt
for ch in default 1 2 4 8; do
  if [ $ch = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS; else export NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=$ch; fi
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_w1_ch$ch.json 2> $out/bench_w1_ch$ch.err; echo "w1 ch=$ch rc=$?" | tee -a $out/rc.txt
done
unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS
timeout -k 10 300 python3 tools/validate_timing.py 256 > $out/validate_timing.json 2> $out/validate.err; echo "validate rc=$?" | tee -a $out/rc.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c1/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['pipeline']['device_ms_per_block'], d['config']['gates_passed'])
    except Exception as e: print(f, 'ERR', e)
print(open('gpurun_out/c1/validate_timing.json').read())
PY
