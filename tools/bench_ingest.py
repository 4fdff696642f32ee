#!/usr/bin/env python3
"""Scene file -> resident scene, both ways, on the same synthetic text:

  python : Parser.parse_scene_file (objects) -> BSPNode.build_bsp_tree (recursive NumPy) -> scene_pack.pack_scene
  native : rt_scene_parse_file (C++ tokenizer, threads) -> rt_scene_build_tree (GPU)

then rt_create for each, a check that the tables are bit-identical, and one JSON line.

    python tools/bench_ingest.py --grid 1000 [--no-python]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=1000)
    ap.add_argument("--no-python", action="store_true")
    a = ap.parse_args()
    import gen_scene
    import ray_tracing_amd  # noqa: F401
    from ray_tracing_amd import abi, native, scene_pack
    from ray_tracing_amd.render import FastScene
    text = gen_scene.synth_mixed(a.grid, 4321, 5, 10)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
        path = f.name
    out = dict(workload="synth_mixed(grid=%d, seed=4321)" % a.grid, text_MB=round(len(text) / 1e6, 1))
    try:
        native.bsp_build(np.zeros((4, 3)), np.ones((4, 3)))        # library + HIP context up before timing
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            fs = FastScene.from_file(path)
            t1 = time.perf_counter()
            ctx = native.Context(fs.packed)
            t2 = time.perf_counter()
            ctx.close()
            run = dict(parse_ms=fs.timing["parse_ms"], tree_ms=fs.timing["tree_ms"], tree_device_ms=fs.timing["tree_device_ms"],
                       rt_create_ms=(t2 - t1) * 1e3, total_ms=(t2 - t0) * 1e3, threads=fs.packed.info["threads"])
            if best is None or run["total_ms"] < best["total_ms"]:
                best = run
        out["native"] = {k: round(v, 2) for k, v in best.items()}
        out["primitives"] = fs.packed.info["n_finite"]
        out["nodes"] = fs.packed.desc.n_nodes
        if not a.no_python:
            from BSPNode import BSPNode
            from Parser import parse_scene_file
            t0 = time.perf_counter()
            cam, st, _o, surfaces, mats, lights = parse_scene_file(path)
            t1 = time.perf_counter()
            bsp = BSPNode.build_bsp_tree(surfaces)
            t2 = time.perf_counter()
            packed = scene_pack.pack_scene(surfaces, mats, lights, cam, st, bsp)
            t3 = time.perf_counter()
            ctx = native.Context(packed)
            t4 = time.perf_counter()
            ctx.close()
            out["python"] = dict(parse_s=round(t1 - t0, 2), tree_s=round(t2 - t1, 2), pack_s=round(t3 - t2, 2),
                                 rt_create_ms=round((t4 - t3) * 1e3, 1), total_s=round(t4 - t0, 2))
            x, y = fs.packed.arrays, packed.arrays
            out["identical_tables"] = bool(all(np.array_equal(np.asarray(x[k]).reshape(-1), np.asarray(y[k]).reshape(-1))
                                               for k in abi.SCENE_TABLES))
            out["speedup"] = round((t4 - t0) * 1e3 / best["total_ms"], 1)
    finally:
        os.unlink(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
