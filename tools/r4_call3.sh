#!/bin/bash
# round 4, call 3: where the deep-LCP blocks spend their time now (per round, per kernel); the one-rank RCCL line again
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c3; mkdir -p $out
ARCHON_HIP_LIB=$PWD/dark-archon_amd/libarchon_hip_exp.so ARCHON_TRACE_ROUNDS=1 timeout -k 10 200 python3 tools/stage_times.py 256 prose 2 > $out/rounds_prose.txt 2>&1; echo "trace prose rc=$?"
ARCHON_HIP_LIB=$PWD/dark-archon_amd/libarchon_hip_exp.so ARCHON_TRACE_ROUNDS=1 timeout -k 10 300 python3 tools/real_text.py 256 > $out/rounds_real.txt 2>&1; echo "trace real rc=$?"
bash tools/prof_kernels.sh prose > /dev/null 2>&1 && cp gpurun_out/prof_prose.txt $out/kernels_prose.txt; echo "prof prose rc=$?"
bash tools/prof_kernels.sh real > /dev/null 2>&1 && cp gpurun_out/prof_real.txt $out/kernels_real_text.txt; echo "prof real rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_w1.json 2> $out/bench_w1.err; echo "w1 rc=$?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c3/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['pipeline']['device_ms_per_block'], d['config']['gates_passed'])
    except Exception as e: print(f, 'ERR', e)
PY
grep "^round\|general_stage" $out/rounds_prose.txt | head -40
cat $out/kernels_prose.txt
