#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c15; mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_inverse.py -x -q > $out/tests_inv.log 2>&1; rc=$?; echo "inv tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests_inv.log
[ $rc = 0 ] || exit 1
for mb in 256 64 16 4; do
  timeout -k 10 200 python3 tools/stage_times.py $mb random 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/mb=$mb /" | tee -a $out/inv.txt
done
timeout -k 10 200 python3 tools/stage_times.py 256 text 3 inv 2>/dev/null | grep '^inverse' | tail -1 | cut -c1-200 | sed "s/^/text /" | tee -a $out/inv.txt
hipcc -O3 --offload-arch=gfx950 -o /tmp/scatter_pass tools/micro/scatter_pass.hip && timeout -k 10 300 /tmp/scatter_pass | tee $out/micro_scatter_pass.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "validate or resident or container or cli or post" > $out/tests2.log 2>&1; rc=$?; echo "tests2 rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests2.log
