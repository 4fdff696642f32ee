#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection.csv files per kernel and counter: tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/pass*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if filt and filt not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in acc:
    print("==", k)
    for c, v in sorted(acc[k].items()):
        print("  %-28s %18.0f   (%d dispatches)" % (c, v, len(calls[(k, c)])))

# derived: share of the SIMD issue slots (1024 SIMDs, one VALU instruction = one 4-cycle slot; GRBM_GUI_ACTIVE is summed
# over the 8 XCDs) in which a VALU / any instruction was issued while the kernel ran
print("\n%-44s %10s %10s %10s %12s" % ("kernel", "VALU %", "any inst %", "lanes %", "GRBM cycles"))
for k in acc:
    a = acc[k]
    if a.get("GRBM_GUI_ACTIVE", 0) <= 0 or "SQ_ACTIVE_INST_VALU" not in a:
        continue
    slots = a["GRBM_GUI_ACTIVE"] * 32.0
    lanes = a.get("SQ_THREAD_CYCLES_VALU", 0) / max(a.get("SQ_INSTS_VALU", 1), 1) / 64.0 * 100.0
    print("%-44s %10.1f %10.1f %10.1f %12.0f" % (k[:44], 100.0 * a["SQ_ACTIVE_INST_VALU"] / slots,
                                                  100.0 * a.get("SQ_ACTIVE_INST_ANY", 0) / slots, lanes, a["GRBM_GUI_ACTIVE"] / 8.0))
