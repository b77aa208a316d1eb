#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c11; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_forward.py -m gpu -x -q -k "bucket_mode or shapes_vs_oracle or ragged or known or compacted or alphabet" > $out/tests_a.log 2>&1; rc=$?
tail -2 $out/tests_a.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-shapes > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$out/bench_line.json'));print(d['value'],d['ms_per_step'],d['pipeline']['device_ms_per_block'],d['pipeline']['host_us_buffer_forward_submit'],d['roofline']['kernels'])"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
