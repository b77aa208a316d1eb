#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/c8; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_post.py tests/test_gpu_cli.py -m gpu -x -q -k "not config5" > $out/tests_quick.log 2>&1; rc=$?; echo "quick tests rc=$rc" | tee -a $out/rc.txt
tail -15 $out/tests_quick.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python3 tools/post_decode_bench.py 256 2>$out/pd.err | tee $out/post_decode.txt
timeout -k 10 600 python3 tools/config5.py 256 2>/dev/null | tail -1 | tee $out/config5.json
