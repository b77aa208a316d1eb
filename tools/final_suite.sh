#!/bin/bash
# the round's last call: the whole -m gpu suite, then the inverse's evidence on the last library
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-final}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests_final.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -2 $out/tests_final.log
[ $rc = 0 ] || exit 1
bash tools/inv_kernels.sh 8 > $out/inverse_kernels.txt 2>&1; cat $out/inverse_kernels.txt | cut -c1-150
for mib in 256 128 64 16 4; do
  timeout -k 10 120 python3 tools/stage_times.py $mib random 3 inv 2>/dev/null | tail -1 | sed "s/^/inverse-random-${mib}MiB /" | tee -a $out/stage_times_inverse.txt | cut -c1-200
done
