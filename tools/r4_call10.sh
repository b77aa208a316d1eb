#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/c10; mkdir -p $out
L=$PWD/dark-archon_amd
for rep in 1 2; do
bash tools/ab_real.sh "$L/libarchon_hip.so $L/libarchon_hip_fu512.so $L/libarchon_hip_fu256.so $L/libarchon_hip_fu128.so" 2>&1 | tee -a $out/ab_fu.txt
done
for l in libarchon_hip.so libarchon_hip_fu256.so; do
  for sh in text random_copy motif_defects dna; do
    ARCHON_HIP_LIB=$L/$l timeout -k 10 200 python3 tools/stage_times.py 256 $sh 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l $sh', d['ms_total'])" | tee -a $out/ab_fu_other.txt
  done
done
