set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for args in "--width 1920 --height 1080" "--workload synth1M" "--workload synth90k" "--workload pool" "--workload original --width 500 --height 500"; do
  echo "== $args"
  timeout -k 10 200 python tools/prof_frame.py --frames 8 --no-profile $args 2>&1 | grep ms_total
done
