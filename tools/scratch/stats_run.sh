#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export RT_AMD_LIB=$PWD/ray-tracing_amd/csrc/librt_amd_stats.so
timeout -k 10 120 python tools/trav_stats.py --width 1920 --height 1080 2>&1 | grep -v "^\[" | head -40 || exit 1
timeout -k 10 120 python tools/trav_stats.py --workload original --width 500 --height 500 2>&1 | grep -v "^\[" | head -60
