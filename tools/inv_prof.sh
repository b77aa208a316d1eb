#!/bin/bash
# inverse on a 256 MiB random block: chain granularity sweep, then kernel-level times (rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
for sb in 8 7 6 5; do
  echo "== sbits $sb"
  ARCHON_INV_SBITS=$sb timeout -k 10 120 python3 tools/stage_times.py 256 random 3 inv 2>/dev/null | tail -1 || exit 1
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/invprof -- python3 tools/stage_times.py 256 random 3 inv > gpurun_out/invprof.log 2>&1
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/invprof/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'inv::' in r['Name'] or 'hist256' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
