#!/usr/bin/env python3
"""rocprofv3 counter passes (tools/pmc.sh + tools/pmc_traffic.sh output dirs) -> one JSON per workload with, per kernel
and per FRAME: executed floating-point operations, VALU issue share, active lanes, HBM-side bytes.

    python tools/pmc_roofline.py <pmc dir> <traffic dir or -> <frames> <tree> <out.json> [note]

executed flop = (ADD + MUL + TRANS + 2 FMA, float32 and float64) x 64 lanes x (SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU)):
the counters count wave-level instructions, the lane factor is the kernel's average share of active lanes.
HBM bytes = 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB-ish units of 1024 B... see
the guide: FETCH_SIZE / WRITE_SIZE are reported in kilobytes."""
import collections
import csv
import glob
import json
import sys


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in sorted(glob.glob(d + "/pass*/*/*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    return acc, disp


def main():
    pmc_dir, traffic_dir, frames, tree, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    acc, disp = load(pmc_dir)
    tacc = load(traffic_dir)[0] if traffic_dir != "-" else {}
    kernels = {}
    for k, a in acc.items():
        if "SQ_INSTS_VALU" not in a or a["SQ_INSTS_VALU"] <= 0:
            continue
        lanes = a.get("SQ_THREAD_CYCLES_VALU", 0.0) / a["SQ_INSTS_VALU"] / 64.0
        f32 = a.get("SQ_INSTS_VALU_ADD_F32", 0) + a.get("SQ_INSTS_VALU_MUL_F32", 0) + a.get("SQ_INSTS_VALU_TRANS_F32", 0) + 2 * a.get("SQ_INSTS_VALU_FMA_F32", 0)
        f64 = a.get("SQ_INSTS_VALU_ADD_F64", 0) + a.get("SQ_INSTS_VALU_MUL_F64", 0) + a.get("SQ_INSTS_VALU_TRANS_F64", 0) + 2 * a.get("SQ_INSTS_VALU_FMA_F64", 0)
        e = {
            "dispatches_per_frame": len(disp[(k, "SQ_INSTS_VALU")]) / frames,
            "valu_wave_instructions_per_frame": a["SQ_INSTS_VALU"] / frames,
            "salu_wave_instructions_per_frame": a.get("SQ_INSTS_SALU", 0) / frames,
            "fp32_wave_instructions_per_frame": {n: a.get("SQ_INSTS_VALU_%s_F32" % n, 0) / frames for n in ("ADD", "MUL", "FMA", "TRANS")},
            "fp64_wave_instructions_per_frame": {n: a.get("SQ_INSTS_VALU_%s_F64" % n, 0) / frames for n in ("ADD", "MUL", "FMA", "TRANS")},
            "int32_wave_instructions_per_frame": a.get("SQ_INSTS_VALU_INT32", 0) / frames,
            "lanes_active": round(lanes, 4),
            "executed_flop_per_frame": (f32 + f64) * 64.0 * lanes / frames,
        }
        if a.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_ACTIVE_INST_VALU" in a:
            slots = a["GRBM_GUI_ACTIVE"] * 32.0          # GRBM_GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs, one VALU instruction = a 4-cycle slot
            e["valu_issue_share"] = round(a["SQ_ACTIVE_INST_VALU"] / slots, 4)
            e["any_instruction_issue_share"] = round(a.get("SQ_ACTIVE_INST_ANY", 0) / slots, 4)
        t = tacc.get(k)
        if t and ("FETCH_SIZE" in t or "WRITE_SIZE" in t):
            e["hbm_bytes_per_frame"] = (2.0 * t.get("FETCH_SIZE", 0) + t.get("WRITE_SIZE", 0)) * 1024.0 / frames
            e["fetch_size_kb_per_frame"] = t.get("FETCH_SIZE", 0) / frames
            e["write_size_kb_per_frame"] = t.get("WRITE_SIZE", 0) / frames
        kernels[k] = e
    json.dump({"what": __doc__.split("\n\n")[0], "frames_in_run": frames, "tree": tree, "note": note, "kernels": kernels},
              open(out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1]["valu_wave_instructions_per_frame"])[:12]:
        print("%-34s valu %7.1f M  flop %7.2f G  lanes %.2f  valu share %s  hbm %s MB" % (
            k[:34], e["valu_wave_instructions_per_frame"] / 1e6, e["executed_flop_per_frame"] / 1e9, e["lanes_active"],
            e.get("valu_issue_share"), round(e.get("hbm_bytes_per_frame", 0) / 1e6, 1) if "hbm_bytes_per_frame" in e else None))


if __name__ == "__main__":
    main()
