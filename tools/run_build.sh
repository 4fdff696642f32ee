set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python tools/bench_ingest.py --grid 1000 > gpurun_out/ingest_1M.json 2> gpurun_out/ingest_1M.err && cat gpurun_out/ingest_1M.json
timeout -k 10 300 python bench.py > gpurun_out/bench_k.json 2> gpurun_out/bench_k.err && cat gpurun_out/bench_k.json
