#!/bin/bash
# experiment: sparse waves / more deeper-level workgroups per CU for k_intersect of the deeper levels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
run() {  # env..., then args
  echo "== $*"
  env "$@" 2>/dev/null | grep "ms_total per" 
}
for cfg in "RT_AMD_SPARSE=0 RT_AMD_SIDE_HIT=1" "RT_AMD_SPARSE=0 RT_AMD_SIDE_HIT=2" "RT_AMD_SPARSE=0 RT_AMD_SIDE_HIT=4" "RT_AMD_SPARSE=2 RT_AMD_SIDE_HIT=1" "RT_AMD_SPARSE=2 RT_AMD_SIDE_HIT=2" "RT_AMD_SPARSE=2 RT_AMD_SIDE_HIT=4" "RT_AMD_SPARSE=3 RT_AMD_SIDE_HIT=4"; do
  for args in "--workload original --width 500 --height 500" "--width 3840 --height 2160 --world 8 --rank 0" "--width 1920 --height 1080" "--workload pool"; do
    echo "== $cfg | $args"
    env $cfg timeout -k 10 120 python tools/prof_frame.py --frames 12 --no-profile $args 2>&1 | grep "ms_total per" || exit 1
  done
done
