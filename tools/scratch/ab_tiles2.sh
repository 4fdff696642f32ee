#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for args in "--width 3840 --height 2160 --world 8 --rank 0" "--width 1280 --height 720" "--width 640 --height 360"; do
  for t in 1500000 400000 100000; do
    export RT_AMD_TILES_MIN=$t
    echo -n "tiles_min=$t sync'd "; timeout -k 10 200 python tools/prof_frame.py --frames 12 --no-profile $args 2>&1 | grep "ms_total per" | cut -c1-140 || exit 1
  done
done
for t in 1500000 400000 100000; do export RT_AMD_TILES_MIN=$t; echo -n "tiles_min=$t "; timeout -k 10 200 python tools/b2b.py --width 640 --height 360 2>&1 | grep "^b2b" | cut -c1-120; done
