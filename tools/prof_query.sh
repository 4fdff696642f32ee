#!/bin/bash
# usage: tools/prof_query.sh <tag> [bench_query args]  - wall times, then kernel times of the two batched queries
tag=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python tools/bench_query.py "$@" > gpurun_out/$tag/query.json 2> gpurun_out/$tag/query.err || { echo "query bench failed"; tail -5 gpurun_out/$tag/query.err; exit 1; }
cat gpurun_out/$tag/query.json
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/prof/*/*kernel_stats.csv" % sys.argv[1])[0]
for row in csv.DictReader(open(f)):
    if "k_visibility" in row["Name"] or "k_intersect_only" in row["Name"]:
        print("%-40s calls %4s  avg %10.1f us" % (row["Name"].split("(")[0][:40], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
