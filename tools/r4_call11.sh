#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c11; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $out/rc.txt
tail -3 $out/tests.log
[ $rc = 0 ] || exit 1
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
