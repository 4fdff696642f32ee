"""Random scene files with unusual proportions (tiny / huge lights, lights inside the field or grazing a plane, touching and
overlapping primitives, 0-3 planes): the default path against the float64 reference-order walk, bit for bit.
    python tools/fuzz_scenes.py [--n 60] [--seed 1] [--flight]
--flight: besides, six frames of every scene enqueued back to back on one stream (other seeds, two frame sizes) against the
same frames rendered alone with their levels in sequence (the early camera pass, rt_amd.hip: AltL0)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracing_amd import abi, native  # noqa: E402
from ray_tracing_amd.render import FastScene  # noqa: E402


def make_scene(rng):
    L = []
    ext = float(10 ** rng.uniform(0.0, 2.0))                      # field half size 1 .. 100
    n = int(10 ** rng.uniform(0.0, 3.5))
    cam_d = ext * rng.uniform(1.2, 3.0)
    L.append("cam %r %r %r 0 0 0 0 1 0 %r 1" % (float(rng.uniform(-0.3, 0.3) * cam_d), float(rng.uniform(0.2, 0.9) * cam_d),
                                               float(-cam_d), float(rng.uniform(0.8, 2.0))))
    sh = int(rng.choice([3, 3, 4, 5, 7, 11, 16, 17, 2, 1]))
    L.append("set %r %r %r %d %d" % (0.2, 0.3, 0.5, sh, int(rng.integers(0, 7))))
    n_mtl = 6
    for m in range(n_mtl):
        kd, ks = rng.uniform(0.1, 0.9, 3), rng.uniform(0, 1, 3)
        kr = rng.uniform(0, 0.5, 3) if m % 3 == 0 else np.zeros(3)
        tr = float(rng.uniform(0.2, 0.9)) if m % 3 == 1 else 0.0
        L.append("mtl %s %s %s %d %r" % (" ".join(repr(float(x)) for x in kd), " ".join(repr(float(x)) for x in ks),
                                         " ".join(repr(float(x)) for x in kr), int(rng.integers(1, 40)), tr))
    n_pl = int(rng.integers(0, 4))
    ground = -ext * 0.1
    if n_pl >= 1:
        L.append("pln 0 1 0 %r %d" % (ground, 1 + int(rng.integers(0, n_mtl))))
    for _ in range(max(0, n_pl - 1)):
        nv = rng.normal(size=3); nv[1] = abs(nv[1])
        L.append("pln %r %r %r %r %d" % (float(nv[0]), float(nv[1]), float(nv[2]), float(-ext * rng.uniform(0.5, 2.0)), 1 + int(rng.integers(0, n_mtl))))
    size0 = ext / max(n, 1) ** (1 / 3.0)
    box_frac = rng.choice([0.0, 0.3, 0.7, 1.0])
    for i in range(n):
        c = rng.uniform(-ext, ext, 3); c[1] = rng.uniform(ground, ground + ext)
        sz = float(size0 * 10 ** rng.uniform(-1.0, 0.5))
        mode = rng.integers(0, 6)
        if rng.uniform() < box_frac:
            if mode == 0 and n_pl >= 1:
                c[1] = ground + sz / 2                              # resting exactly on the plane (when representable)
            if mode == 1:
                c = np.round(c / sz) * sz                           # lattice: coincident faces between neighbours
            L.append("box %r %r %r %r %d" % (float(c[0]), float(c[1]), float(c[2]), sz, 1 + int(rng.integers(0, n_mtl))))
        else:
            if mode == 0 and n_pl >= 1:
                c[1] = ground + sz                                  # tangent to the plane
            L.append("sph %r %r %r %r %d" % (float(c[0]), float(c[1]), float(c[2]), sz, 1 + int(rng.integers(0, n_mtl))))
    for _ in range(int(rng.integers(1, 5))):
        kind = rng.integers(0, 5)
        if kind == 0:
            pos = rng.uniform(-ext, ext, 3); pos[1] = ground + ext * rng.uniform(1.5, 5)          # above
        elif kind == 1:
            pos = rng.uniform(-ext, ext, 3); pos[1] = rng.uniform(ground, ground + ext)            # inside the field
        elif kind == 2:
            pos = rng.uniform(-ext, ext, 3) * 3; pos[1] = ground + ext * 10 ** rng.uniform(-4, -1)   # grazing the ground
        elif kind == 3:
            pos = rng.normal(size=3) * ext * 100                                                    # very far
        else:
            pos = np.array([0.0, ground + ext * 2, 0.0]) + rng.choice([0.0, 1.0], 3) * rng.uniform(-ext, ext, 3)   # axis aligned offsets
        radius = float(ext * 10 ** rng.uniform(-3.0, 0.7))
        shadow = float(rng.choice([0.0, 0.5, 0.9, 1.0], p=[0.1, 0.3, 0.4, 0.2]))
        L.append("lgt %r %r %r 1 1 1 %r %r %r" % (float(pos[0]), float(pos[1]), float(pos[2]), float(rng.uniform(0, 1)), shadow, radius))
    return "\n".join(L) + "\n", sh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dump", default="")
    ap.add_argument("--flight", action="store_true")
    a = ap.parse_args()
    total = 0
    early = 0
    if a.flight:
        import torch
        stream = torch.cuda.Stream()
    for i in range(a.n):
        rng = np.random.default_rng(a.seed * 1000 + i)
        txt, sh = make_scene(rng)
        try:
            sc = FastScene(text=txt)
        except Exception as e:                                    # scenes the tree builder refuses are not the subject here
            print("scene %d: build failed: %s" % (i, e), flush=True)
            continue
        ctx = native.Context(sc.packed, device=0)
        w, h = 200, 120
        p = dict(jitter=True, seed=i)
        fast = ctx.render(abi.render_params(w, h, **p))
        exact = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **p))
        nobeam = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_NO_BEAM, **p))
        d1 = int((np.abs(fast["rgb"].astype(np.float64) - exact["rgb"].astype(np.float64)).max(-1) > 0).sum())
        d2 = int((np.abs(nobeam["rgb"].astype(np.float64) - exact["rgb"].astype(np.float64)).max(-1) > 0).sum())
        di = int((fast["ids"] != exact["ids"]).sum())
        bad = d1 + d2 + di
        if a.flight:
            for (fw, fh) in ((200, 120), (331, 187)):
                n = fw * fh
                e0 = native.lib().rt_debug_early_frames(ctx._h)
                outs = [torch.full((n * 3,), -1.0, dtype=torch.float32, device="cuda:0") for _ in range(6)]
                with torch.cuda.stream(stream):
                    for k in range(6):
                        ctx.render_device(abi.render_params(fw, fh, jitter=True, seed=900 + k), d_rgb=outs[k].data_ptr(), stream=stream.cuda_stream)
                try:
                    ctx.sync_stats()
                except native.NativeError as e:      # queues grew: once more
                    with torch.cuda.stream(stream):
                        for k in range(6):
                            ctx.render_device(abi.render_params(fw, fh, jitter=True, seed=900 + k), d_rgb=outs[k].data_ptr(), stream=stream.cuda_stream)
                    ctx.sync_stats()
                torch.cuda.synchronize()
                early += native.lib().rt_debug_early_frames(ctx._h) - e0
                if ctx.voided_frames(reset=True):
                    continue                             # (a voided frame wrote nothing: not the subject here)
                for k in range(6):
                    ref = torch.full((n * 3,), -2.0, dtype=torch.float32, device="cuda:0")
                    ctx.render_device(abi.render_params(fw, fh, jitter=True, seed=900 + k, flags=abi.RT_FLAG_NO_OVERLAP), d_rgb=ref.data_ptr())
                    ctx.sync_stats()
                    if not torch.equal(outs[k], ref):
                        bad += 1
                        print("   frame in flight %d (%dx%d) differs from the frame alone" % (k, fw, fh), flush=True)
        total += bad
        d = sc.packed.desc
        print("scene %3d: prims %5d planes %d lights %d s %d | fast!=exact %d  nobeam!=exact %d  ids %d  rays %d%s" % (
            i, d.n_spheres + d.n_boxes, d.n_planes, d.n_lights, sh, d1, d2, di, fast["stats"]["rays"], "   <<<<<<" if bad else ""), flush=True)
        if bad and a.dump:
            with open(os.path.join(a.dump, "fuzz_%d_%d.txt" % (a.seed, i)), "w") as f:
                f.write(txt)
        ctx.close()
    if a.flight:
        print("frames that took the early camera pass:", early)
    print("TOTAL:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
