#!/bin/bash
# launch-by-launch device timeline of ONE forward: tools/timeline.sh <real|shape> [MiB] -> gpurun_out/timeline_<tag>.txt
# (rocprofv3 --kernel-trace of two forwards; the second one is listed: start offset, duration, kernel, grid)
export TMPDIR=/tmp
tag=$1; mib=${2:-256}
if [ "$tag" = real ]; then prog="tools/real_text.py $mib"; else prog="tools/stage_times.py $mib $tag 2"; fi
rm -rf gpurun_out/tl_$tag; timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -- python3 $prog > gpurun_out/tl_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv,glob,sys
tag=sys.argv[1]
f=sorted(glob.glob('gpurun_out/tl_%s/*/*_kernel_trace.csv' % tag))[-1]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
# forwards start with the two-byte count (k_hist16) or the byte count
starts=[i for i,r in enumerate(rows) if 'k_hist16<' in r['Kernel_Name'] or 'k_hist256' in r['Kernel_Name']]
firsts=[starts[0]]
for i in starts[1:]:
    if int(rows[i]['Start_Timestamp'])-int(rows[firsts[-1]]['Start_Timestamp']) > 5_000_000: firsts.append(i)
a=firsts[-1]
b=len(rows)
t0=int(rows[a]['Start_Timestamp'])
out=open('gpurun_out/timeline_%s.txt' % tag,'w')
busy=0
for r in rows[a:b]:
    s=int(r['Start_Timestamp'])-t0; d=int(r['End_Timestamp'])-int(r['Start_Timestamp']); busy+=d
    name=r['Kernel_Name'].replace('archon::','').replace('void ','')
    name=name.split('(')[0][:44]
    out.write("%10.3f ms %9.1f us  %-44s grid %s\n" % (s/1e6, d/1e3, name, r.get('Grid_Size_X', r.get('Grid_Size',''))))
out.write("kernels busy %.3f ms of span %.3f ms\n" % (busy/1e6, (int(rows[b-1]['End_Timestamp'])-t0)/1e6))
print("timeline", tag, len(rows[a:b]), "launches, busy %.3f ms" % (busy/1e6))
PY
