#!/bin/bash
# per-round trace of the general stage (experiments library) on one shape: tools/trace_shape.sh <shape> [MiB]
export ARCHON_HIP_LIB=$PWD/dark-archon_amd/libarchon_hip_exp.so ARCHON_TRACE_ROUNDS=1
timeout -k 10 300 python3 tools/stage_times.py ${2:-256} $1 2 2>&1 | tail -${3:-40} | cut -c1-700
