#!/usr/bin/env python3
"""Times the tree build of a synthetic scene: the recursive host restatement of BSPNode.build_bsp_tree
(ray-tracing_amd/host/BSPNode.py, NumPy) beside rt_bsp_build (csrc/rt_build.hip) on the same corner arrays,
checks that both give the same arrays, and prints one JSON line.

    python tools/bench_build.py --grid 1000 [--reps 5] [--no-host]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def corners_of(text):
    """Enclosing corners straight from the scene text, with the expressions of Sphere.py / Cube.py
    (get_enclosing_values): position -/+ radius, position -/+ scale / 2; list order = file order."""
    small, big = [], []
    for line in text.splitlines():
        if line.startswith("sph") or line.startswith("box"):
            v = [float(x) for x in line.split()[1:5]]
            half = v[3] if line.startswith("sph") else v[3] / 2
            small.append([v[0] - half, v[1] - half, v[2] - half])
            big.append([v[0] + half, v[1] + half, v[2] + half])
    return np.array(small), np.array(big)


def host_build(small, big):
    """The host mirror's builder on bare corner arrays (objects only carry get_enclosing_values)."""
    import ray_tracing_amd  # noqa: F401
    from BSPNode import BSPNode
    from surfaces.Object3D import Object3D

    class Boxed(Object3D):
        def __init__(self, k):
            self.k = k

        def get_enclosing_values(self):
            return small[self.k], big[self.k]

    objs = [Boxed(k) for k in range(len(small))]
    t0 = time.perf_counter()
    root = BSPNode.build_bsp_tree(objs)
    return root.flat, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-host", action="store_true")
    a = ap.parse_args()
    import gen_scene
    from ray_tracing_amd import native
    small, big = corners_of(gen_scene.synth_mixed(a.grid, 4321, 5, 10))
    native.bsp_build(small[:64], big[:64])      # loads the library, creates the HIP context
    runs = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        t = native.bsp_build(small, big)
        wall = time.perf_counter() - t0
        runs.append(dict(wall_ms=wall * 1e3, **t["stats"]))
    best = min(runs, key=lambda r: r["ms_total"])
    out = dict(workload="synth_mixed(grid=%d, seed=4321)" % a.grid, primitives=len(small), nodes=best["n_nodes"],
               leaf_refs=best["n_leaf_refs"], depth=best["depth"], launches=best["launches"],
               ref_visits=best["ref_visits"], gpu_ms_device=round(best["ms_device"], 3),
               gpu_ms_build_call=round(best["ms_total"], 3), gpu_ms_with_fetch=round(best["wall_ms"], 3),
               gpu_ms_device_all=[round(r["ms_device"], 3) for r in runs], scratch_MB=round(best["bytes_scratch"] / 1e6, 1))
    # ~72 B of streams per reference and level: classify 4+4+16 r, 8 w; scan 8 r + 8 w; scatter 4+4+8 r, <= 16 w
    out["algorithmic_GB"] = round(72e-9 * best["ref_visits"], 3)
    out["achieved_GBps_device"] = round(72e-9 * best["ref_visits"] / (best["ms_device"] * 1e-3), 1)
    if not a.no_host:
        flat, secs = host_build(small, big)
        same = all(np.array_equal(t[k], getattr(flat, k)) for k in ("lo", "hi", "right", "first", "count", "leaf_prim"))
        out.update(host_numpy_s=round(secs, 3), identical=bool(same), speedup_vs_host=round(secs * 1e3 / best["wall_ms"], 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
