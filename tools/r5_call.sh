#!/bin/bash
export TMPDIR=/tmp
ARCHON_HIP_LIB=$PWD/dark-archon_amd/libarchon_hip_exp.so ARCHON_TRACE_HOST=1 timeout -k 10 120 python3 tools/stage_times.py 256 random 6 2>&1 | grep "host phases" | tail -4
