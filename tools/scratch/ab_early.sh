#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for args in "--width 1920 --height 1080" "--width 3840 --height 2160 --frames 30" "--width 3840 --height 2160 --world 8 --rank 0" "--workload synth1M --frames 10" "--workload pool" "--workload original --width 500 --height 500" "--workload synth90k --frames 20"; do
  for e in 1 0; do
    if [ $e = 1 ]; then export RT_AMD_NO_EARLY_CAM=1; else unset RT_AMD_NO_EARLY_CAM; fi
    echo -n "no_early=$e "; timeout -k 10 200 python tools/b2b.py $args 2>&1 | grep "^b2b" || exit 1
  done
done
