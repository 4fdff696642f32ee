#!/bin/bash
# usage: tools/pmc2.sh <outdir-under-gpurun_out> <program args...>   - issue / LDS / branch counters (second set)
set -e
out=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$out
i=0
for grp in \
 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_BUSY_CU_CYCLES" \
 "SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU2 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$out/pass$i -- python tools/prof_frame.py "$@" > gpurun_out/$out/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
