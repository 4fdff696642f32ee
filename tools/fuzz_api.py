"""Random scenes (tools/fuzz_scenes.make_scene) through every entry point against the oracle: rt_intersect and
rt_visibility on random and degenerate ray batches, rt_render at odd sizes and tile partitions, rt_trace.
    python tools/fuzz_api.py [--n 30] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
from fuzz_scenes import make_scene  # noqa: E402
from ray_tracing_amd import abi, native  # noqa: E402
from ray_tracing_amd.render import FastScene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    total = 0
    for i in range(a.n):
        rng = np.random.default_rng(a.seed * 7919 + i)
        txt, sh = make_scene(rng)
        try:
            sc = FastScene(text=txt)
        except Exception:
            continue
        d = sc.packed.desc
        ctx = native.Context(sc.packed, device=0)
        lo = np.array([d.node_lo[k] for k in range(3)]) if d.n_nodes else -np.ones(3)
        hi = np.array([d.node_hi[k] for k in range(3)]) if d.n_nodes else np.ones(3)
        ext = float(np.abs(np.concatenate([lo, hi])).max()) + 1.0
        n = 4000
        o = rng.uniform(-1.5 * ext, 1.5 * ext, (n, 3))
        dr = rng.normal(size=(n, 3)); dr /= np.linalg.norm(dr, axis=-1, keepdims=True)
        k = np.arange(n)
        dr[k % 9 == 1] = np.eye(3)[rng.integers(0, 3, (k % 9 == 1).sum())] * rng.choice([-1.0, 1.0], ((k % 9 == 1).sum(), 1))
        dr[k % 97 == 2] = 0.0
        dr[k % 97 == 3, 0] = np.nan
        dr[k % 11 == 4] *= 3.7
        o[k % 97 == 5] *= 1e9
        pts, ids = ctx.intersect(o, dr)
        opts, oids, _ = orc.intersect(sc.packed, o, dr)
        e_int = int((ids != oids).sum()) + int((~((pts == opts) | (np.isnan(pts) & np.isnan(opts)))).any(-1).sum())
        tg = opts.copy(); tg[::3] = opts[rng.permutation(n)][::3]; tg[1::7] += 7e-7
        with np.errstate(invalid="ignore"):
            want = np.linalg.norm(opts - tg, axis=-1) < 1e-6
        e_vis = int((ctx.visibility(o, dr, tg) != want).sum())
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        world = int(rng.choice([1, 1, 2, 3, 5])); rank = int(rng.integers(0, world))
        p = abi.render_params(w, h, jitter=bool(rng.integers(0, 2)), seed=i, rank=rank, world=world)
        out = ctx.render(p)
        ref = orc.render(sc.packed, p, threads=16, want_f64=False)
        e_ren = int((out["ids"] != ref["ids"]).sum()) + int((np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max(-1) > 1).sum())
        e_ren += 0 if all(out["stats"][c] == ref["counters"][c] for c in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits")) else 1
        m = 600
        tp = abi.render_params(1, 1, sh_rays=min(sh, 4), rec_max=3, jitter=True, seed=i)
        rgb, st = ctx.trace(tp, o[:m], dr[:m])
        rref, cnt = orc.trace(sc.packed, tp, o[:m], dr[:m])
        e_tr = int((~np.isclose(rgb, rref, rtol=1e-9, atol=1e-12, equal_nan=True)).any(-1).sum()) + (0 if st["rays"] == cnt["rays"] else 1)
        bad = e_int + e_vis + e_ren + e_tr
        total += bad
        print("scene %3d: prims %5d planes %d lights %d | intersect %d visibility %d render(%dx%d r%d/%d) %d trace %d%s" % (
            i, d.n_spheres + d.n_boxes, d.n_planes, d.n_lights, e_int, e_vis, w, h, rank, world, e_ren, e_tr, "   <<<<<<" if bad else ""), flush=True)
        ctx.close()
    print("TOTAL:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
