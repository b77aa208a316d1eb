#!/bin/bash
# the whole -m gpu suite, its tail kept (tools/final_suite.sh does the same behind the bench lines and profiles)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/${1:-suite}; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests_final.log 2>&1; rc=$?
tail -3 $out/tests_final.log
exit $rc
