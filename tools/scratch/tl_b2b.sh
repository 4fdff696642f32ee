#!/bin/bash
# kernel timeline of the last two frames of a back-to-back run
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=$1; shift
rm -rf gpurun_out/tlb_$tag
timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlb_$tag -- python tools/b2b.py --frames 6 "$@" > gpurun_out/tlb_$tag.log 2>&1 || { echo "failed"; tail -5 gpurun_out/tlb_$tag.log; exit 1; }
python - "$tag" <<'PY'
import csv, glob, sys, re
f = glob.glob("gpurun_out/tlb_%s/*/*kernel_trace.csv" % sys.argv[1])[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
gens = [i for i, r in enumerate(rows) if re.search(r"k_intersect<\w+, true", r["Kernel_Name"])]
rows = rows[gens[-2]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f  q%-3s %s" % (s / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"].split("(")[0][:34]))
PY
