#!/bin/bash
# usage: tools/prof_bench.sh <tag>  - the default bench line, then rocprofv3 --kernel-trace --stats of the same workload
# (bench.py --no-cpu-baseline --no-extra: without the CPU sample and without the other scenes' frames of extra_frames, so that
# the per-kernel averages are those of the headline frame alone)
tag=$1
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout -k 10 400 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err || { echo "bench failed"; tail -5 gpurun_out/$tag/bench.err; exit 1; }
cat gpurun_out/$tag/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python bench.py --no-cpu-baseline --no-extra > gpurun_out/$tag/bench_prof.json 2> gpurun_out/$tag/prof.err || { echo "rocprof failed"; tail -5 gpurun_out/$tag/prof.err; exit 1; }
cat gpurun_out/$tag/bench_prof.json
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/prof/*/*kernel_stats.csv" % sys.argv[1])[0]
for row in csv.DictReader(open(f)):
    print("%-60s calls %6s  total %10.3f ms  avg %10.1f us  %5s%%" % (row["Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
