#!/bin/bash
# usage: tools/ab_stage.sh "<lib.so> ..." [shape] [MiB] [reps] -- stage times of several builds of the library, alternating, on one box
libs=$1; shape=${2:-random}; mib=${3:-256}; reps=${4:-3}
for i in $(seq $reps); do
  for l in $libs; do
    echo "== $l"; ARCHON_HIP_LIB=$l python3 tools/stage_times.py $mib $shape 3 2>/dev/null | tail -1
  done
done
