#!/bin/bash
# a fuzz campaign on the round's last library: natural-route cases against the oracle, big blocks by LF-consistency + round trip,
# period defects, the post stage
export TMPDIR=/tmp
out=gpurun_out/fuzz; mkdir -p $out
timeout -k 10 380 python3 tools/fuzz_hunt.py 4012 4400 2>&1 | grep -v "^seed .* done" | tail -5 | tee $out/hunt.txt
timeout -k 10 380 python3 tools/fuzz_big.py 506 560 2>&1 | tee $out/big_all.txt | grep -v "ok True" | tail -5 | tee $out/big.txt
timeout -k 10 300 python3 tools/fuzz_defects.py 130 120 2>&1 | tail -2 | tee $out/defects.txt
timeout -k 10 200 python3 tools/fuzz_post.py 2>&1 | tail -2 | tee $out/post.txt
