set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 200 python tools/bench_build.py --grid 1000 --no-host
