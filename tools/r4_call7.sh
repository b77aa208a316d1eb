#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c7; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
for sh in prose motif_defects; do
  timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | sed "s/^/$sh /" | tee -a $out/stage_times.txt
done
timeout -k 10 300 python3 tools/real_text.py 256 2>/dev/null | tail -1 | tee $out/real_text.json
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_w1.json 2> $out/bench_w1.err; echo "w1 rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c7/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['pipeline']['device_ms_per_block'], d['pipeline']['host_us_buffer_forward_submit'])
PY
