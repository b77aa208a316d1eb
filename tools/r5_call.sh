#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_c4; mkdir -p $out
timeout -k 10 600 python3 tools/stagger_sweep.py 256 2>&1 | tee $out/stagger_sweep.txt | grep groups
