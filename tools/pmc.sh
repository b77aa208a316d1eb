#!/bin/bash
# usage: tools/pmc.sh <outdir> <counter-set-name> <counters...> -- <python args>
# Runs one rocprofv3 --pmc pass of tools/stage_times.py (separate passes per counter set,
# as MI355X_MICROARCH.md prescribes) and leaves the CSVs under gpurun_out/<outdir>/<set>.
# Guard: a set that asks one hardware block for more counters than it has slots makes rocprofv3 abort inside its
# finaliser and hang (round 1, gpurun_out/call31): at most 8 SQ_*, 4 TCC_* (FETCH_SIZE counts 3, WRITE_SIZE 2),
# 4 of TCP_* / TA_* each, 2 GRBM_* per pass.
out=$1; shift; set=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
nsq=0; ntcc=0; ntcp=0; nta=0; ngrbm=0
for c in "${ctrs[@]}"; do
  case $c in
    SQ_*) nsq=$((nsq+1));; FETCH_SIZE) ntcc=$((ntcc+3));; WRITE_SIZE) ntcc=$((ntcc+2));; TCC_*) ntcc=$((ntcc+1));;
    TCP_*) ntcp=$((ntcp+1));; TA_*) nta=$((nta+1));; GRBM_*) ngrbm=$((ngrbm+1));;
  esac
done
if [ $nsq -gt 8 ] || [ $ntcc -gt 4 ] || [ $ntcp -gt 4 ] || [ $nta -gt 4 ] || [ $ngrbm -gt 2 ]; then
  echo "pmc.sh: counter set '$set' exceeds the per-block slots (SQ $nsq/8, TCC $ntcc/4, TCP $ntcp/4, TA $nta/4, GRBM $ngrbm/2): split it" >&2
  exit 2
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out; timeout -k 10 150 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/$out/$set -- python3 tools/stage_times.py "$@" > gpurun_out/$out.$set.log 2>&1
