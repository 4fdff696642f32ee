#!/bin/bash
# A/B: per-kernel times (levels in sequence) of the HEAD library and the working tree's
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export GRAFT_REPO_ROOT=$PWD
for lib in head new; do
  if [ $lib = head ]; then export RT_AMD_LIB=$PWD/ray-tracing_amd/csrc/librt_amd_head.so; else unset RT_AMD_LIB; fi
  bash tools/kt.sh ${lib}_10k --width 1920 --height 1080 || exit 1
  bash tools/kt.sh ${lib}_orig --workload original --width 500 --height 500 || exit 1
done
