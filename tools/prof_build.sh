#!/bin/bash
# usage: tools/prof_build.sh <tag>  - rocprofv3 --kernel-trace --stats of the GPU tree build (1M primitives, 5 builds)
tag=$1
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python tools/bench_build.py --grid 1000 --no-host > gpurun_out/$tag/build.json 2> gpurun_out/$tag/prof.err || { echo "rocprof failed"; tail -5 gpurun_out/$tag/prof.err; exit 1; }
cat gpurun_out/$tag/build.json
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/prof/*/*kernel_stats.csv" % sys.argv[1])[0]
print(f)
for row in csv.DictReader(open(f)):
    print("%-40s calls %6s  total %10.3f ms  avg %10.1f us  %5s%%" % (row["Name"].replace("(anonymous namespace)::", "").split("(")[0][:40], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY
