#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for args in "--width 3840 --height 2160 --world 8 --rank 0" "--width 1280 --height 720" "--width 960 --height 540"; do
  for t in 1500000 400000; do
    export RT_AMD_TILES_MIN=$t
    echo -n "tiles_min=$t "; timeout -k 10 200 python tools/b2b.py $args 2>&1 | grep "^b2b" || exit 1
  done
done
