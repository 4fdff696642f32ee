#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <program args...>
# One rocprofv3 --pmc pass per counter group (the tool does not split passes itself); each pass has
# its own hard time limit; a failing pass ends the script.
set -e
out=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$out
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
 "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VMEM" \
 "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCC_HIT TCC_MISS GRBM_GUI_ACTIVE" \
 "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$out/pass$i -- python tools/prof_frame.py "$@" > gpurun_out/$out/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
