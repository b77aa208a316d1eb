#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_fuzz; mkdir -p $out
timeout -k 10 120 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | tail -1 | cut -c1-200
timeout -k 10 300 python3 tools/fuzz_hunt.py 5000 5250 2>&1 | grep -v "^seed .* done" | tail -5 | tee $out/hunt.txt
ARCHON_ALIGNED_MIN=65536 timeout -k 10 200 python3 tools/fuzz_hunt.py 5250 5400 2>&1 | grep -v "^seed .* done" | tail -3 | tee $out/hunt_bucket_mode.txt
timeout -k 10 420 python3 tools/fuzz_big.py 600 660 2>&1 | tee $out/big_all.txt | grep -v "ok True" | tail -5 | tee $out/big.txt
grep -c "ok True" $out/big_all.txt
timeout -k 10 200 python3 tools/fuzz_defects.py 300 100 2>&1 | tail -2 | tee $out/defects.txt
