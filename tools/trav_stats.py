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
        for l in range(12):
            r = buf[8 * l]
            if not r:
                continue
            print(l, r, *["%.2f" % (buf[8 * l + k] / r) for k in range(1, 5)], buf[8 * l + 5], buf[8 * l + 6], buf[8 * l + 7])
        print({k: v for k, v in out["stats"].items() if k.startswith("ms_")})
        if hasattr(lib, "rt_debug_trav_time"):
            rc = lib.rt_debug_trav_time(buf, 8 * 64)
            assert rc == 0
            print("level: longest wave of k_intersect, us")
            for l in range(12):
                if buf[8 * l + 7]:
                    print(l, "%.1f" % (buf[8 * l + 7] / 100.0))


if __name__ == "__main__":
    main()
