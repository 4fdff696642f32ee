#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir-under-gpurun_out> <prof_frame args...>
# HBM-side traffic per kernel (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in separate passes.
set -e
out=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 400 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$out/pass$i -- python tools/prof_frame.py "$@" > gpurun_out/$out/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py gpurun_out/$out
