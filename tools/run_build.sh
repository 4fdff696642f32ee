set -e
cd /root/repo
timeout -k 10 300 python -u -m pytest tests/test_gpu_build.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/build_tests.log 2>&1 || { tail -40 gpurun_out/build_tests.log; exit 1; }
tail -3 gpurun_out/build_tests.log
timeout -k 10 120 python tools/bench_build.py --grid 300 > gpurun_out/build_90k.json 2> gpurun_out/build_90k.err && cat gpurun_out/build_90k.json
timeout -k 10 200 python tools/bench_build.py --grid 1000 > gpurun_out/build_1M.json 2> gpurun_out/build_1M.err && cat gpurun_out/build_1M.json
