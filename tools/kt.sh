#!/bin/bash
# usage: tools/kt.sh <tag> <prof_frame args...>   - rocprofv3 kernel trace of a few frames, prints per-kernel ms per frame
tag=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kt_$tag
timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$tag -- python tools/prof_frame.py --frames 3 "$@" > gpurun_out/kt_$tag.log 2>&1 || { echo "kt $tag failed"; tail -5 gpurun_out/kt_$tag.log; exit 1; }
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/kt_%s/*/*kernel_stats.csv" % sys.argv[1])[0]
tot = 0
out = []
for row in csv.DictReader(open(f)):
    ms = float(row["TotalDurationNs"]) / 1e6 / 3
    tot += ms
    if ms > 0.02: out.append("%s %.2f" % (row["Name"].split("(")[0].replace("void ", "")[:18], ms))
print("[%s] total %.2f ms/frame | " % (sys.argv[1], tot) + " | ".join(out))
PY
