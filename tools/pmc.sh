#!/bin/bash
# usage: tools/pmc.sh <outdir> <counter-set-name> <counters...> -- <python args>
# Runs one rocprofv3 --pmc pass of tools/stage_times.py (separate passes per counter set,
# as MI355X_MICROARCH.md prescribes) and leaves the CSVs under gpurun_out/<outdir>/<set>.
out=$1; shift; set=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out; timeout -k 10 150 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/$out/$set -- python3 tools/stage_times.py "$@" > gpurun_out/$out.$set.log 2>&1
