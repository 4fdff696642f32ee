#!/usr/bin/env python3
"""Per-level traversal counters of the nearest-hit query (diagnostics build: make -C ray-tracing_amd/csrc stats).
usage: RT_AMD_LIB=ray-tracing_amd/csrc/librt_amd_stats.so python tools/trav_stats.py [prof_frame-style args]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth10k")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-rays", type=int, default=None)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1)
    a = ap.parse_args()
    import bench
    from ray_tracing_amd import abi, native
    scene, _ = bench.build_scene(a.workload)
    lib = native.lib()
    with native.Context(scene.packed) as ctx:
        # shadows off the float64 path: the counters then hold the nearest-hit queries of the levels only (RT_FLAG_PROFILE: levels in sequence)
        out = ctx.render(abi.render_params(a.width, a.height, sh_rays=a.sh_rays, rank=a.rank, world=a.world, seed=0, flags=abi.RT_FLAG_PROFILE), want_rgb=False, want_ids=False)
        buf = (C.c_ulonglong * (8 * 64))()
        rc = lib.rt_debug_trav_stats(buf, 8 * 64)
        assert rc == 0
        print("level rays node/ray leaf/ray batch/ray conf/ray max_node max_leaf max_conf")
        rays = {}
        for l in range(12):
            r = buf[8 * l]
            rays[l] = r
            if not r:
                continue
            print(l, r, *["%.2f" % (buf[8 * l + k] / r) for k in range(1, 5)], buf[8 * l + 5], buf[8 * l + 6], buf[8 * l + 7])
        print({k: v for k, v in out["stats"].items() if k.startswith("ms_")})
        if hasattr(lib, "rt_debug_trav_time"):
            rc = lib.rt_debug_trav_time(buf, 8 * 64)
            assert rc == 0
            wave_log(lib)
            print("level: longest wave of k_intersect, us | clocks per ray (as seen by the ray's lane, waiting for its wave included): descents, float32 leaf tests, float64 confirmations")
            for l in range(12):
                if buf[8 * l + 7]:
                    r = max(rays.get(l, 1), 1)
                    print(l, "%.1f" % (buf[8 * l + 7] / 100.0), "|", *["%.0f" % (buf[8 * l + k] / r) for k in range(3)], "| exact-walk rays %d (%d ties), clocks each %.0f" % (buf[8 * l + 4], buf[8 * l + 5], buf[8 * l + 3] / max(buf[8 * l + 4], 1)))


def wave_log(lib):
    """the slowest waves of every level: whole wave, inside nearest-hit calls, busiest lane's descents / leaf tests / confirmations / reference-order walk (clocks)"""
    if not hasattr(lib, "rt_debug_wave_log"):
        return
    buf = (C.c_ulonglong * (16 * 4096 * 6))()
    cnt = (C.c_uint32 * 16)()
    assert lib.rt_debug_wave_log(buf, cnt) == 0
    print("level: waves | median wave clocks | the three slowest waves: total, in nearest-hit calls, busiest lane: descents, leaf tests, confirmations, exact walk")
    for l in range(16):
        n = min(cnt[l], 4096)
        if not n:
            continue
        rows = sorted((tuple(buf[(l * 4096 + i) * 6 + k] for k in range(6)) for i in range(n)), reverse=True)
        print(l, n, "|", rows[n // 2][0], "|", *["%d/%d/%d+%d+%d+%d" % r for r in rows[:3]])


if __name__ == "__main__":
    main()
