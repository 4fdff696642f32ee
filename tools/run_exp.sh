set -e
cd /root/repo
export RT_AMD_WALK_MIN=1
rm -rf gpurun_out/pmc_walk gpurun_out/pmc_walk2
bash tools/pmc.sh pmc_walk --frames 1 --no-profile --workload original --width 500 --height 500 > gpurun_out/pmc_walk.log 2>&1 || { tail gpurun_out/pmc_walk.log; exit 1; }
bash tools/pmc2.sh pmc_walk2 --frames 1 --no-profile --workload original --width 500 --height 500 >> gpurun_out/pmc_walk.log 2>&1 || true
python tools/pmc_summary.py gpurun_out/pmc_walk "k_shadow_walk"
python tools/pmc_summary.py gpurun_out/pmc_walk2 "k_shadow_walk"
