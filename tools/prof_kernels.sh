#!/bin/bash
# per-kernel device time of the forward pipeline: tools/prof_kernels.sh <real|shape> [MiB]  -> gpurun_out/prof_<tag>.txt
# (rocprofv3 --kernel-trace --stats; the python program itself sits behind "--")
export TMPDIR=/tmp
tag=$1; mib=${2:-256}
if [ "$tag" = real ]; then prog="tools/real_text.py $mib"; else prog="tools/stage_times.py $mib $tag 3"; fi
rm -rf gpurun_out/prof_$tag; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 $prog > gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv,glob,sys
tag=sys.argv[1]
f=sorted(glob.glob('gpurun_out/prof_%s/*/*_kernel_stats.csv' % tag))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
out=open('gpurun_out/prof_%s.txt' % tag,'w')
for r in rows:
    if float(r['Percentage']) > 0.3:
        line="%-72s calls %5s  total %9.3f ms  avg %9.1f us  %5.2f %%" % (r['Name'][:72], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage']))
        print(line); out.write(line+"\n")
PY
