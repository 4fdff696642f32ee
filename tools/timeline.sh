#!/bin/bash
# usage: tools/timeline.sh <tag> <prof_frame args...>  - kernel timeline (start, duration, name) of the LAST frame
tag=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl_$tag
timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -- python tools/prof_frame.py --frames 3 --no-profile "$@" > gpurun_out/tl_$tag.log 2>&1 || { echo "timeline $tag failed"; tail -5 gpurun_out/tl_$tag.log; exit 1; }
python - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/tl_%s/*/*kernel_trace.csv" % sys.argv[1])[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
import re
gens = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gen") or re.search(r"k_intersect<\w+, true", r["Kernel_Name"]) or "k_small_level<true>" in r["Kernel_Name"]]
rows = rows[gens[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0][:32]))
PY
