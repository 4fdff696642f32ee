#!/bin/bash
# usage (GPU box): bash tools/pmc_roofline.sh <tag> <out-name> <prof_frame args...>
# All counter passes of tools/pmc.sh + the two HBM passes over 2 profile frames of one workload, then the per-kernel summary
# bench.py reads: gpurun_out/<tag>/<out-name>.json (copy it to profiles/).
set -e
tag=$1; name=$2; shift; shift
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/pmc.sh $tag/pmc --frames 2 "$@" | tail -5
bash tools/pmc_traffic.sh $tag/traffic --frames 2 "$@" > gpurun_out/$tag/traffic.txt 2>&1 || { tail -5 gpurun_out/$tag/traffic.txt; exit 1; }
python tools/pmc_summary.py gpurun_out/$tag/pmc > gpurun_out/$tag/pmc_summary.txt
python tools/pmc_roofline.py gpurun_out/$tag/pmc gpurun_out/$tag/traffic 2 "${RT_TREE:-unknown}" gpurun_out/$tag/$name.json "prof_frame.py --frames 2 $*"
