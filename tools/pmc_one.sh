#!/bin/bash
# usage: tools/pmc_one.sh <tag> <kernel-substring> <prof_frame args...>
# ONE rocprofv3 --pmc pass (instruction mix of the SQ block) over a few frames; prints the per-kernel sums.
tag=$1; filt=$2; shift; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc1_$tag; mkdir -p gpurun_out/pmc1_$tag
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc1_$tag/pass1 -- python tools/prof_frame.py --frames 2 "$@" > gpurun_out/pmc1_$tag/pass1.log 2>&1 || { echo "pmc pass failed"; tail -5 gpurun_out/pmc1_$tag/pass1.log; exit 1; }
echo "[$tag]"; python tools/pmc_summary.py gpurun_out/pmc1_$tag "$filt" | grep -v "^$" | head -40
