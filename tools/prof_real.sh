#!/bin/bash
# kernel-level times of the forward pipeline on real text (tools/real_text.py)
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_real -- python3 tools/real_text.py ${1:-256} > gpurun_out/prof_real.log 2>&1
tail -1 gpurun_out/prof_real.log | cut -c1-400
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_real/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.4: print(r['Name'][:64], r['Calls'], "%.3f ms total" % (float(r['TotalDurationNs'])/1e6), r['Percentage'])
PY
