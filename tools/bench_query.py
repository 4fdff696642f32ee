#!/usr/bin/env python3
"""Batched queries through the C-ABI on one workload: rt_intersect (nearest hit) and rt_visibility (any hit) on the
same shadow-ray batch - the shaded points of a camera pass, one jittered ray per point and light, as
Light.get_shadow_rays makes them.  Prints wall times (host buffers in and out); run under rocprofv3
--kernel-trace --stats for the kernel times.  No torch, no oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth10k")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    import bench
    from ray_tracing_amd import native
    scene, _ = bench.build_scene(a.workload)
    lights = scene.packed.arrays["lgt_pos"].reshape(-1, 3)
    radius = scene.packed.arrays["lgt_radius"]
    rng = np.random.default_rng(0)
    with native.Context(scene.packed) as ctx:
        src, dirs = ctx.initial_rays(a.width, a.height)
        pts, ids = ctx.intersect(src, dirs)
        P = pts[ids != 0]
        O, D, T = [], [], []
        for m in range(len(lights)):
            s = lights[m] + rng.uniform(-0.5, 0.5, P.shape) * radius[m]
            v = P - s
            O.append(s); D.append(v / (np.linalg.norm(v, axis=-1, keepdims=True) + 1e-6)); T.append(P)
        O, D, T = np.concatenate(O), np.concatenate(D), np.concatenate(T)
        res = dict(workload=a.workload, rays=int(len(O)))
        for name, fn in (("rt_intersect", lambda: ctx.intersect(O, D)), ("rt_visibility", lambda: ctx.visibility(O, D, T))):
            best = 1e9
            for _ in range(a.repeat):
                t0 = time.perf_counter()
                out = fn()
                best = min(best, time.perf_counter() - t0)
            res[name + "_wall_ms"] = round(best * 1e3, 2)
        lit = ctx.visibility(O, D, T)
        pts2, _ = ctx.intersect(O, D)
        with np.errstate(invalid="ignore"):
            res["identical"] = bool(np.array_equal(lit, np.linalg.norm(pts2 - T, axis=-1) < 1e-6))
        res["lit_fraction"] = round(float(lit.mean()), 4)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
