#!/bin/bash
# kernel-level times (rocprofv3 --kernel-trace --stats) of the forward pipeline on one 256 MiB shape
export TMPDIR=/tmp
sh=${1:-dna}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$sh -- python3 tools/stage_times.py 256 $sh 3 > gpurun_out/prof_$sh.log 2>&1
python3 - "$sh" <<'PY'
import csv,glob,sys
f=sorted(glob.glob('gpurun_out/prof_%s/*/*_kernel_stats.csv' % sys.argv[1]))[-1]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.3: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
