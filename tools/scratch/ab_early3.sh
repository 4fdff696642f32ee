#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for args in "--width 3840 --height 2160 --world 2 --rank 1 --frames 30" "--width 2560 --height 1440 --frames 40" "--width 3840 --height 2160 --world 4 --rank 1" "--width 1280 --height 720" "--workload synth400" "--workload synth10kmix"; do
  for e in 1 0; do
    if [ $e = 1 ]; then export RT_AMD_NO_EARLY_CAM=1; else unset RT_AMD_NO_EARLY_CAM; fi
    echo -n "no_early=$e "; timeout -k 10 200 python tools/b2b.py $args 2>&1 | grep "^b2b" || exit 1
  done
done
