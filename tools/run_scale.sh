#!/bin/bash
# usage: bash tools/run_scale.sh - frame time of one rank's share of a tile-partitioned 3840x2160 frame (world 1, 2, 4, 8) on one GPU
set -e
cd /root/repo
for args in "--width 1920 --height 1080" "--width 3840 --height 2160" "--width 3840 --height 2160 --world 2 --rank 1" "--width 3840 --height 2160 --world 4 --rank 1" "--width 3840 --height 2160 --world 8 --rank 0" "--width 3840 --height 2160 --world 8 --rank 5"; do
  echo "== $args"
  timeout -k 10 120 python tools/prof_frame.py --frames 8 --no-profile $args 2>&1 | tail -2
done
