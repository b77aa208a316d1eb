#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r05_c22; mkdir -p $out
t0=$(date +%s)
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python3 -c "
import json;d=json.load(open('$out/bench_default.json'));print(d['value'],d['ms_per_step'],d['steps'],d['config']['gates_passed'],sorted(d.keys()))"
