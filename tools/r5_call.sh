#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_final2; mkdir -p $out
timeout -k 10 400 python3 bench.py --steps 20 --warmup 2 > $out/bench_line.json 2> $out/bench.err || exit 1
echo "bench done"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline 2> $out/bench_w1.err | tail -1 > $out/bench_line_torchrun_world1.json || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-shapes > $out/bench_line_same_box_again.json 2>> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shapes > $out/stats.log 2>&1 || exit 1
cp $(ls $out/stats/*/*_kernel_stats.csv | tail -1) $out/kernel_stats.csv
echo "stats done"
timeout -k 10 120 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | tail -1 | cut -c1-200
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $out/tests_final.log 2>&1; tail -2 $out/tests_final.log
