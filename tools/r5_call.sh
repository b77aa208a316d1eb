#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c15; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_gpu_forward.py tests/test_gpu_inverse.py -m gpu -x -q -k "shapes_vs_oracle or bucket_mode or known or inverse or concurrent or block_coder" > $out/tests_a.log 2>&1; rc=$?
tail -2 $out/tests_a.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-shapes > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$out/bench_line.json'));print(d['value'],d['ms_per_step'],d['pipeline']['device_ms_per_block'],d['pipeline']['host_us_buffer_forward_submit'])"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-shapes 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['pipeline']['device_ms_per_block'],d['pipeline']['host_us_buffer_forward_submit'])"
bash tools/inv_kernels.sh 8 2>&1 | head -4 | cut -c1-150
