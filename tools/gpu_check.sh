#!/bin/bash
# usage (on the GPU box, e.g. through gpurun): bash tools/gpu_check.sh
# The GPU test suite, then the frame time of every workload of bench.py (8 frames each, overlapped schedule).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for args in "--width 1920 --height 1080" "--width 3840 --height 2160" "--width 3840 --height 2160 --world 8 --rank 0" "--workload synth90k" "--workload synth1M" "--workload pool --sh-rays 1" "--workload pool" "--workload original --width 500 --height 500"; do
  echo "== $args"
  timeout -k 10 200 python tools/prof_frame.py --frames 8 --no-profile $args 2>&1 | grep ms_total
done
