#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r05_c21; mkdir -p $out
make host >/dev/null 2>&1
for i in 1 2 3 4 5 6 7 8; do
timeout -k 10 200 python3 -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "block_coder_objects or concurrent" > $out/t$i.log 2>&1; echo "run $i rc=$?"
done
