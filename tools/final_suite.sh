#!/bin/bash
# the round's last call: bench lines of the last library (graded line with shapes / inverse / cpu_baseline, the same again, one rank
# under torch.distributed.run), rocprofv3 kernel stats of the bench command, the inverse's kernels, then the whole -m gpu suite
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-final}; mkdir -p $out
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err || exit 1
echo "bench done"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2> $out/bench_w1.err | tail -1 > $out/bench_line_torchrun_world1.json || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shapes > $out/bench_line_same_box_again.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --in-flight 1 --no-cpu-baseline --no-shapes > $out/bench_line_one_block_at_a_time.json 2>> $out/bench.err || exit 1
# kernel_stats.csv: one block at a time -- the kernels' own durations, what `roofline` quotes; kernel_stats_two_in_flight.csv: the default command
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 1 --in-flight 1 --no-cpu-baseline --no-shapes > $out/stats.log 2>&1 || exit 1
cp $(ls $out/stats/*/*_kernel_stats.csv | tail -1) $out/kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-shapes > $out/stats2.log 2>&1 || exit 1
cp $(ls $out/stats2/*/*_kernel_stats.csv | tail -1) $out/kernel_stats_two_in_flight.csv
echo "stats done"
for sh in random dna text a ab motif prose motif_defects random_copy; do
  timeout -k 10 120 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | sed "s/^/$sh /" >> $out/stage_times.txt || exit 1
done
timeout -k 10 120 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | tail -1 | sed "s/^/inverse-random /" >> $out/stage_times.txt
for mib in 4 16 64 128; do
  timeout -k 10 120 python3 tools/stage_times.py $mib random 3 2>/dev/null | tail -1 | sed "s/^/forward-random-${mib}MiB /" >> $out/stage_times.txt
  timeout -k 10 120 python3 tools/stage_times.py $mib random 3 inv 2>/dev/null | tail -1 | sed "s/^/inverse-random-${mib}MiB /" >> $out/stage_times.txt
done
echo "stage times done"
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $out/tests_final.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -2 $out/tests_final.log
