set -e
cd /root/repo
rm -rf gpurun_out/pmc_gen
bash tools/pmc.sh pmc_gen --frames 1 --no-profile > gpurun_out/pmc_gen.log 2>&1 || { tail gpurun_out/pmc_gen.log; exit 1; }
bash tools/pmc2.sh pmc_gen2 --frames 1 --no-profile >> gpurun_out/pmc_gen.log 2>&1 || true
python tools/pmc_summary.py gpurun_out/pmc_gen "k_intersect<true, true>"
python tools/pmc_summary.py gpurun_out/pmc_gen2 "k_intersect<true, true>"
