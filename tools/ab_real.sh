#!/bin/bash
# usage: tools/ab_real.sh "<lib.so> ..." -- prose stage times and the source-text corpus with several builds of the library on one box
for l in $1; do
  echo "== $l"
  ARCHON_HIP_LIB=$l python3 tools/stage_times.py 256 prose 2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('prose', d['ms_total'], d['doubling_rounds'])"
  ARCHON_HIP_LIB=$l python3 tools/real_text.py 256 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('real', round(d['ms_total'],2), d['doubling_rounds'])"
done
