#!/bin/bash
# per-kernel device times of the inverse on a 256 MiB random block, by rows per chain head (ARCHON_INV_SBITS): tools/inv_kernels.sh [sbits ...]
export TMPDIR=/tmp
for sb in "$@"; do
  rm -rf gpurun_out/invk_$sb
  ARCHON_INV_SBITS=$sb timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/invk_$sb -- python3 tools/stage_times.py 256 random 3 inv > gpurun_out/invk_$sb.log 2>&1 || exit 1
  echo "== sbits $sb: $(grep '^inverse' gpurun_out/invk_$sb.log | tail -1 | cut -c1-160)"
  python3 - $sb <<'PY'
import csv,glob,sys
f=sorted(glob.glob('gpurun_out/invk_%s/*/*_kernel_stats.csv' % sys.argv[1]))[-1]
for r in csv.DictReader(open(f)):
    if 'inv::' in r['Name'] or 'hist' in r['Name']: print("   %-50s calls %4s avg %9.1f us total %8.3f ms" % (r['Name'].replace('archon::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
