#!/bin/bash
# a fuzz campaign on the round's last library: natural-route cases against the oracle (product routing, then bucket mode with
# range-relative records forced on small blocks), big blocks by LF-consistency + round trip, period defects
# usage: tools/fuzz_suite.sh <tag> <first seed>
export TMPDIR=/tmp
out=gpurun_out/${1:-fuzz}; s0=${2:-7000}; mkdir -p $out
timeout -k 10 300 python3 tools/fuzz_hunt.py $s0 $((s0+250)) 2>&1 | grep -v "^seed .* done" | tail -5 | tee $out/hunt.txt
ARCHON_SMALL_BLOCK=0 ARCHON_ALIGNED_MIN=65536 ARCHON_REL_MIN_SEG=1 timeout -k 10 300 python3 tools/fuzz_hunt.py $((s0+250)) $((s0+450)) 2>&1 | grep -v "^seed .* done" | tail -3 | tee $out/hunt_bucket_mode.txt
timeout -k 10 420 python3 tools/fuzz_big.py $((s0+1000)) $((s0+1060)) 2>&1 | tee $out/big_all.txt | grep -v "ok True" | tail -5 | tee $out/big.txt
echo "big blocks ok: $(grep -c 'ok True' $out/big_all.txt)" | tee -a $out/big.txt
timeout -k 10 200 python3 tools/fuzz_defects.py $((s0+2000)) 100 2>&1 | tail -2 | tee $out/defects.txt
timeout -k 10 200 python3 tools/stress_objects.py 60 2>&1 | tail -2 | tee $out/stress_objects.txt
ARCHON_SMALL_BLOCK=0 STRESS_SYNC_ROUTES=1 timeout -k 10 200 python3 tools/stress_objects.py 60 2>&1 | tail -2 | tee -a $out/stress_objects.txt
timeout -k 10 300 python3 tools/stress_in_flight.py 400 2 256 2>&1 | tail -2 | tee $out/stress_in_flight.txt
timeout -k 10 300 python3 tools/stress_in_flight.py 300 3 128 2>&1 | tail -2 | tee -a $out/stress_in_flight.txt
