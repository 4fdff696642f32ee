#!/usr/bin/env python3
"""Render a few frames of one workload through the C-ABI - the small program put after `--` in
rocprofv3 runs (kernel trace or one --pmc pass).  No torch, no oracle."""
import argparse
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
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--sh-rays", type=int, default=None)
    ap.add_argument("--rec-max", type=int, default=None)
    ap.add_argument("--no-jitter", action="store_true")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1, help="render only the tiles of one rank of a tile-partitioned frame")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernel families with events (keeps the overlapped schedule)")
    a = ap.parse_args()
    import bench
    from ray_tracing_amd import abi, native
    scene, _ = bench.build_scene(a.workload)
    with native.Context(scene.packed) as ctx:
        ms = []
        for k in range(a.frames):
            out = ctx.render(abi.render_params(a.width, a.height, sh_rays=a.sh_rays, rec_max=a.rec_max, rank=a.rank, world=a.world,
                                               jitter=not a.no_jitter, seed=k, flags=a.flags | (0 if a.no_profile else abi.RT_FLAG_PROFILE)),
                             want_rgb=False, want_ids=False)
            ms.append(round(out["stats"]["ms_total"], 3))
        st = out["stats"]
        print("ms_total per frame:", ms, "| frame state %.2f GB" % (ctx.frame_bytes() / 1e9))
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})


if __name__ == "__main__":
    main()
