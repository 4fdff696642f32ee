#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  Only its OUTPUTS (tests/golden/*.npz, small) are used by tests.
The reference is imported unmodified with the harness-side shims SURVEY.md §8c
lists:

  * `cv2` is absent           -> empty stub module (only `display_image` uses it)
  * `np.NaN` removed in NumPy 2 -> alias to np.nan (Sphere.py:83)
  * bare-name imports         -> /root/reference on sys.path, no bytecode written

Two variants of every BSP-dependent golden are produced:
  "ref"    the reference exactly as it behaves;
  "q4fix"  the same code with `np.any(<int index array>)` in BSPNode.py:156
           evaluated as "array is non-empty" (the reference tests
           index VALUES, so a node whose only entering ray is local ray 0 is
           skipped - batch-order dependent, not reproducible per ray).  This is
           done by giving the BSPNode module a proxy for `np`; no reference file
           is edited.  The number of Q4 events is recorded beside each golden.

Usage:  python tools/make_golden.py [--only NAME,NAME...] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
if not hasattr(np, "NaN"):
    np.NaN = np.nan
sys.path.insert(0, REF)
sys.path.insert(1, HERE)

import warnings  # noqa: E402

warnings.filterwarnings("ignore")

import BSPNode as ref_bsp_mod  # noqa: E402
import Light as ref_light_mod  # noqa: E402
import Material as ref_material_mod  # noqa: E402
import ray_functions as ref_rf  # noqa: E402
import ray_tracer as ref_rt  # noqa: E402
from Parser import parse_scene_file  # noqa: E402
from SceneSettings import SceneSettings  # noqa: E402
from surfaces.Cube import Cube  # noqa: E402
from surfaces.InfinitePlane import InfinitePlane  # noqa: E402
from surfaces.Sphere import Sphere  # noqa: E402
from surfaces.SurfaceAbs import get_surfaces_normals  # noqa: E402
import util as ref_util  # noqa: E402

import gen_scene  # noqa: E402

BSPNode = ref_bsp_mod.BSPNode


# --------------------------------------------------------------------------- helpers
class _NpQ4Proxy:
    """Stands in for `np` inside BSPNode.py only; counts/fixes Q4."""

    def __init__(self, fix: bool):
        self.fix = fix
        self.events = 0

    def __getattr__(self, name):
        return getattr(np, name)

    def any(self, a, *args, **kw):
        if not args and not kw and isinstance(a, np.ndarray) and a.ndim == 1 and a.dtype.kind == "i":
            if a.size > 0 and not np.any(a):
                self.events += 1
                if self.fix:
                    return True
            return bool(np.any(a))
        return np.any(a, *args, **kw)


class q4_mode:
    def __init__(self, fix: bool):
        self.proxy = _NpQ4Proxy(fix)

    def __enter__(self):
        ref_bsp_mod.np = self.proxy
        return self.proxy

    def __exit__(self, *exc):
        ref_bsp_mod.np = np


class no_jitter:
    """np.random.uniform -> zeros (the only RNG call sites are Light.py:277-278)."""

    def __enter__(self):
        self.orig = np.random.uniform
        np.random.uniform = lambda lo, hi, size=None: np.zeros(size)

    def __exit__(self, *exc):
        np.random.uniform = self.orig


class count_rays:
    """Counts rays handed to BSPNode.intersect_vectorize (the Mrays/s numerator)."""

    def __enter__(self):
        self.calls = 0
        self.rays = 0
        self.orig = BSPNode.intersect_vectorize
        outer = self

        def wrapped(node, ray_sources, ray_directions):
            outer.calls += 1
            outer.rays += int(ray_sources.shape[0])
            return outer.orig(node, ray_sources=ray_sources, ray_directions=ray_directions)

        BSPNode.intersect_vectorize = wrapped
        return self

    def __exit__(self, *exc):
        BSPNode.intersect_vectorize = self.orig


def load_scene(path):
    camera, settings, _objs, surfaces, materials, lights = parse_scene_file(path)
    bsp = BSPNode.build_bsp_tree(surfaces=surfaces)
    return camera, settings, surfaces, materials, lights, bsp


def quantize(img, h, w):
    """save_image's quantisation (util.py:94-98) without touching the disk."""
    return (img.reshape((h, w, 3)) * 255).astype(np.uint8)


def render(scene_path, w, h, sh=None, rec=None, jitter=False, seed=None, fix_q4=False):
    camera, settings, surfaces, materials, lights, bsp = load_scene(scene_path)
    settings = SceneSettings(settings.background_color,
                             settings.root_number_shadow_rays if sh is None else sh,
                             settings.max_recursions if rec is None else rec)
    src, dirs = ref_rf.get_initial_rays(camera, image_width=w, image_height=h)
    t0 = time.time()
    if seed is not None:
        np.random.seed(seed)
    with q4_mode(fix_q4) as q4, count_rays() as cnt:
        if jitter:
            img = ref_rt.ray_tracing(src, dirs, surfaces, materials, lights, settings, camera, bsp)
        else:
            with no_jitter():
                img = ref_rt.ray_tracing(src, dirs, surfaces, materials, lights, settings, camera, bsp)
    dt = time.time() - t0
    meta = dict(width=w, height=h, sh_rays=settings.root_number_shadow_rays, rec_max=settings.max_recursions,
                jitter=bool(jitter), seed=seed, q4_fixed=bool(fix_q4), q4_events=q4.events,
                intersect_calls=cnt.calls, rays=cnt.rays, seconds=round(dt, 2))
    return img, meta


def dump_tree(root):
    """Pre-order walk of the reference tree: bounds, leaf flags, leaf surface ids."""
    lo, hi, is_leaf, first, count, leaf_ids = [], [], [], [], [], []

    def rec(n):
        lo.append(np.asarray(n.small_bound, dtype=np.float64))
        hi.append(np.asarray(n.big_bound, dtype=np.float64))
        if n.surfaces is None:
            is_leaf.append(0)
            first.append(-1)
            count.append(0)
            rec(n.left)
            rec(n.right)
        else:
            is_leaf.append(1)
            first.append(len(leaf_ids))
            count.append(len(n.surfaces))
            leaf_ids.extend(int(s.surface.index) for s in n.surfaces)

    rec(root)
    planes = [int(p.index) for p in (root.inf_plains or [])]
    return dict(lo=np.array(lo), hi=np.array(hi), is_leaf=np.array(is_leaf, dtype=np.int8),
                first=np.array(first, dtype=np.int32), count=np.array(count, dtype=np.int32),
                leaf_ids=np.array(leaf_ids, dtype=np.int32), root_planes=np.array(planes, dtype=np.int32))


def save(out_dir, name, meta=None, **arrays):
    path = os.path.join(out_dir, name + ".npz")
    if meta is not None:
        arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)
    print("  wrote %-44s %8.1f KB  %s" % (name + ".npz", os.path.getsize(path) / 1024, meta or ""), flush=True)


# --------------------------------------------------------------------------- goldens
def g_rays(out, scenes):
    for tag, w, h in (("pool", 16, 9), ("pool", 7, 5), ("original", 12, 12), ("small", 9, 6)):
        cam = load_scene(scenes[tag])[0]
        src, dirs = ref_rf.get_initial_rays(cam, image_width=w, image_height=h)
        save(out, "rays_%s_%dx%d" % (tag, w, h), dict(width=w, height=h), sources=src, directions=dirs,
             cam_x=cam.x_dir, cam_y=cam.y_dir, cam_z=cam.z_dir)


def _edge_rays(rng, n):
    o = rng.uniform(-3, 3, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    # hand-made edge cases: axis-parallel rays, zero components (incl. -0.0), origins on slab planes
    extra_o = np.array([[0, 0, -5], [0, 0, 0], [0.5, 0.5, -5], [0.5, 0, -5], [1, 1, -5], [0, 5, 0], [0, 0.25, -5],
                        [0.5, 0.5, 0.5], [0.5, 0.2, -3], [-0.5, -0.5, -5], [2, 0, 0], [0, 0, 5], [0, 1, -5.],
                        [0, 1.0000001, -5], [0, 0.9999999, -5], [0.2, 0.1, 0.05]], dtype=np.float64)
    extra_d = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, -1, 0], [0, -0.0, 1],
                        [1, 0, 0], [0, 0, 1], [0, 0, 1], [-1, 0, 0], [0, 0, -1], [0, 0, 1.],
                        [0, 0, 1], [0, 0, 1], [0.6, 0.0, 0.8]], dtype=np.float64)
    return np.vstack([o, extra_o]), np.vstack([d, extra_d])


def g_prims(out, scenes):
    rng = np.random.default_rng(11)
    o, d = _edge_rays(rng, 3000)
    # rays aimed at the primitives so roughly half of them hit
    tgt = rng.uniform(-1.2, 1.2, (1500, 3))
    o2 = rng.uniform(-4, 4, (1500, 3))
    d2 = tgt - o2
    d2 /= np.linalg.norm(d2, axis=-1, keepdims=True)
    o = np.vstack([o, o2])
    d = np.vstack([d, d2])
    sph = Sphere([0.1, -0.2, 0.3], 1.0, 1, 1)
    cub = Cube(np.array([0.0, 0.0, 0.0]), 1.0, 1, 2)
    cub2 = Cube(np.array([0.3, -0.07, 1.0]), 0.13, 1, 3)
    pln = InfinitePlane([0.0, 1.0, -0.25], -0.55, 1, 4)
    with np.errstate(all="ignore"):
        hs = sph.intersect_vectorized(o, d)
        hc = cub.intersect_vectorized(o, d)
        hc2 = cub2.intersect_vectorized(o, d)
        hp = pln.intersect_vectorized(o, d)
        ms = ~np.isnan(hs[:, 0])
        mc = ~np.isnan(hc[:, 0])
        ns = np.full_like(hs, np.nan)
        ns[ms] = sph.calculate_normals(hs[ms])
        nc = np.full_like(hc, np.nan)
        nc[mc] = cub.calculate_normals(hc[mc])
        # cube normals also on hand-made face/edge/corner points (first-wins ties)
        pts = np.array([[0.5, 0, 0], [0.5, 0.5, 0], [0.5, 0.5, 0.5], [-0.5, 0.5, 0], [0, 0.5, 0.5], [0, -0.5, -0.5],
                        [0.5, 0.49, 0.2], [0.1, 0.2, -0.5], [-0.5, -0.5, -0.5], [0.3, 0.5, 0.1]], dtype=np.float64)
        ncp = cub.calculate_normals(pts)
        nodemask_in, _ = Cube.cube_intersection_mask(o, d, np.array([-0.5, -0.3, 0.2]), np.array([0.5, 0.7, 3.8]),
                                                     outside_hits_only=False)
        nodemask_out, tn = Cube.cube_intersection_mask(o, d, np.array([-0.5, -0.3, 0.2]), np.array([0.5, 0.7, 3.8]),
                                                       outside_hits_only=True)
    save(out, "prims", None, origins=o, directions=d,
         sphere_c=sph.position, sphere_r=np.float64(sph.radius), sphere_hits=hs, sphere_normals=ns,
         cube_c=cub.position, cube_s=np.float64(cub.scale), cube_hits=hc, cube_normals=nc,
         cube2_c=cub2.position, cube2_s=np.float64(cub2.scale), cube2_hits=hc2,
         cube_pts=pts, cube_pts_normals=ncp,
         plane_n_in=np.array([0.0, 1.0, -0.25]), plane_off_in=np.float64(-0.55),
         plane_n=pln.normal, plane_d=np.float64(pln.offset), plane_hits=hp,
         box_lo=np.array([-0.5, -0.3, 0.2]), box_hi=np.array([0.5, 0.7, 3.8]),
         node_mask=nodemask_in[:, 0], prim_mask=nodemask_out[:, 0], t_near=tn[:, 0])


def g_bsp(out, scenes):
    for tag in ("pool", "original", "small", "synth400"):
        cam, st, surfaces, mats, lights, bsp = load_scene(scenes[tag])
        d = dump_tree(bsp)
        save(out, "bsp_" + tag, dict(nodes=int(len(d["is_leaf"])), leaves=int(d["is_leaf"].sum()),
                                     leaf_refs=int(len(d["leaf_ids"]))), **d)


def g_primary(out, scenes):
    for tag, w, h in (("pool", 1920, 1080), ("original", 500, 500), ("small", 320, 200), ("synth400", 320, 180)):
        cam, st, surfaces, mats, lights, bsp = load_scene(scenes[tag])
        src, dirs = ref_rf.get_initial_rays(cam, image_width=w, image_height=h)
        t0 = time.time()
        with q4_mode(False) as q4:
            pts, ids = bsp.intersect_vectorize(ray_sources=src, ray_directions=dirs)
        ev = q4.events
        with q4_mode(True):
            pts_f, ids_f = bsp.intersect_vectorize(ray_sources=src, ray_directions=dirs)
        stride = max(1, (w * h) // 4096)
        save(out, "primary_%s_%dx%d" % (tag, w, h),
             dict(width=w, height=h, q4_events=ev, ids_differ_q4=int((ids != ids_f).sum()),
                  seconds=round(time.time() - t0, 2)),
             ids=ids.astype(np.int16 if len(surfaces) < 32000 else np.int32),
             ids_q4fix=ids_f.astype(np.int16 if len(surfaces) < 32000 else np.int32),
             sample_index=np.arange(0, w * h, stride, dtype=np.int64), sample_points=pts_f[::stride])


def g_secondary(out, scenes):
    """intersect_vectorize on incoherent rays that start inside the scene volume."""
    rng = np.random.default_rng(5)
    for tag, n in (("original", 6000), ("small", 6000), ("pool", 4000), ("synth400", 4000)):
        cam, st, surfaces, mats, lights, bsp = load_scene(scenes[tag])
        lo, hi = np.asarray(bsp.small_bound, float), np.asarray(bsp.big_bound, float)
        pad = 0.25 * (hi - lo)
        o = rng.uniform(lo - pad, hi + pad, (n, 3))
        d = rng.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        with q4_mode(True):
            pts, ids = bsp.intersect_vectorize(ray_sources=o, ray_directions=d)
        save(out, "secondary_" + tag, dict(n=n), origins=o, directions=d, points=pts, ids=ids.astype(np.int32))


def g_shadow_grid(out, scenes):
    rng = np.random.default_rng(3)
    n = 40
    hits = rng.uniform(-2, 2, (n, 3))
    light = ref_light_mod.Light([0.5, 3.0, -1.0], [1, 1, 1], 1.0, 0.9, 1.5, 9)
    hits[0] = [0.5, -1.0, -1.0]            # straight below: D = (0,-1,0), Dz == 0 (Q7)
    hits[1] = [2.0, 1.0, -1.0]             # Dz == 0, generic
    hits[2] = [0.5, 3.0, 2.0]              # Dx == Dy == 0
    arrays = dict(hits=hits, light_pos=light.position, light_radius=np.float64(light.radius))
    L = ref_light_mod.compute_light_rays(hits, [light])[0]
    arrays["light_dirs"] = L
    for s in (1, 2, 5):
        with no_jitter():
            src, vec = ref_light_mod.get_shadow_rays(light, s, -L, hits)
        arrays["src_s%d" % s] = src
        arrays["dir_s%d" % s] = vec
    save(out, "shadow_grid", None, **arrays)


def g_visibility(out, scenes):
    """The lit mask of Light.compute_shadows (Light.py:206-221) on jittered shadow rays of real hit points."""
    for tag, w, h, s in (("pool", 32, 18, 3), ("original", 40, 40, 3), ("small", 40, 25, 3), ("synth400", 40, 22, 2)):
        cam, st, surfaces, mats, lights, bsp = load_scene(scenes[tag])
        src, dirs = ref_rf.get_initial_rays(cam, image_width=w, image_height=h)
        with q4_mode(True):
            pts, ids = bsp.intersect_vectorize(ray_sources=src, ray_directions=dirs)
        hits = pts[ids != 0]
        np.random.seed(77)
        O, D, T, M = [], [], [], []
        for light in lights:
            if light.shadow_intensity == 0:
                continue
            L = ref_light_mod.compute_light_rays(hits, [light])[0]
            ssrc, svec = ref_light_mod.get_shadow_rays(light, s, -L, hits)
            n = hits.shape[0]
            ssrc = ssrc.reshape((n * s * s, 3))
            svec = svec.reshape((n * s * s, 3))
            with q4_mode(True):
                lh, _ = bsp.intersect_vectorize(ray_sources=ssrc, ray_directions=svec)
            tg = np.tile(hits[:, np.newaxis, np.newaxis, :], (1, s, s, 1)).reshape((n * s * s, 3))
            lit = np.linalg.norm(lh - tg, axis=-1) < ref_light_mod.EPSILON
            O.append(ssrc); D.append(svec); T.append(np.repeat(np.arange(n, dtype=np.int32), s * s)); M.append(lit)
            assert np.array_equal(hits[T[-1]], tg)
        O, D, T, M = np.concatenate(O), np.concatenate(D), np.concatenate(T), np.concatenate(M)
        save(out, "visibility_" + tag, dict(width=w, height=h, sh_rays=s, rays=int(len(M)), lit=int(M.sum())),
             origins=O, directions=D, hits=hits, target_index=T, lit=M.astype(np.uint8))


def g_shade_terms(out, scenes):
    rng = np.random.default_rng(4)
    n, Lc = 500, 3
    N = rng.normal(size=(n, 3))
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    N[:50] *= rng.uniform(0.5, 1.5, (50, 1))       # non-unit normals are re-normalised in specular only
    Ld = rng.normal(size=(Lc, n, 3))
    Ld /= np.linalg.norm(Ld, axis=-1, keepdims=True)
    V = rng.normal(size=(n, 3))
    V /= np.linalg.norm(V, axis=-1, keepdims=True)
    Kd = rng.uniform(0, 1, (n, 3))
    Ks = rng.uniform(0, 1, (n, 3))
    Imd = rng.uniform(0, 1, (Lc, n, 3))
    Ims = rng.uniform(0, 1, (Lc, n, 3))
    alpha = rng.integers(0, 60, n)
    alpha[:5] = [0, 1, 2, 30, 31]
    diff = ref_light_mod.compute_diffuse_color(Kd, Imd, Ld, N)
    with np.errstate(all="ignore"):
        spec = ref_light_mod.compute_specular_colors(Ks, alpha, Ld, V, N, Ims)
    mats = [ref_material_mod.Material([0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0, 0, 0], 2.7, 0.0, 0),
            ref_material_mod.Material([0.9, 0.8, 0.7], [0.6, 0.5, 0.4], [0.1, 0, 0], 30.0, 0.25, 1),
            ref_material_mod.Material([0.5, 0.5, 0.5], [1, 1, 1], [0, 0, 1e-3], 0.99, 1.0, 2)]
    mi = rng.integers(0, 3, 64)
    kd2, ks2, ph2, kr2, tr2 = ref_material_mod.get_materials_base_colors(mats, mi)
    save(out, "shade_terms", None, N=N, L=Ld, V=V, Kd=Kd, Ks=Ks, Imd=Imd, Ims=Ims, alpha=alpha, diffuse=diff,
         specular=spec, mat_index=mi, mat_kd=kd2, mat_ks=ks2, mat_phong=ph2, mat_kr=kr2, mat_tr=tr2,
         mat_shininess_in=np.array([2.7, 30.0, 0.99]), mat_transparency_in=np.array([0.0, 0.25, 1.0]))


FRAMES = [
    # name, scene, w, h, sh, rec
    ("pool_64x36_s1", "pool", 64, 36, 1, None),
    ("pool_160x90_s5", "pool", 160, 90, None, None),
    ("pool_1920x1080_s1", "pool", 1920, 1080, 1, None),
    ("original_100x100", "original", 100, 100, None, None),
    ("original_500x500", "original", 500, 500, None, None),
    ("small_160x100", "small", 160, 100, None, None),
    ("small_64x40_s3_r6", "small", 64, 40, 3, 6),
    ("synth400_160x90_s2_r3", "synth400", 160, 90, 2, 3),
]


def g_frames(out, scenes, which=None, frames=None):
    for name, tag, w, h, sh, rec in (frames or FRAMES):
        if which and name not in which:
            continue
        img, meta = render(scenes[tag], w, h, sh=sh, rec=rec, jitter=False, fix_q4=False)
        img_f, meta_f = render(scenes[tag], w, h, sh=sh, rec=rec, jitter=False, fix_q4=True)
        meta["scene"] = tag
        meta["rays_q4fix"] = meta_f["rays"]
        meta["q4_pixels_differ_8bit"] = int(np.any(quantize(img.clip(0, 1), h, w) != quantize(img_f.clip(0, 1), h, w),
                                                    axis=-1).sum())
        arrays = dict(rgb8=quantize(img.clip(0, 1), h, w), rgb8_q4fix=quantize(img_f.clip(0, 1), h, w))
        if w * h <= 20000:
            arrays["rgb_f64_q4fix"] = img_f          # unclipped float64, small frames only
        save(out, "frame_" + name + "_nojitter", meta, **arrays)


def g_oncut(out, scenes):
    """Rays with an exactly zero direction component whose origin lies exactly on a node bound (the reference's
    0 * inf = NaN cull, Cube.py:120-136): crafted batches through intersect_vectorize, primary hit ids and
    deterministic frames of the mirror-symmetric scene at odd widths."""
    cam, st, surfaces, mats, lights, bsp = load_scene(scenes["oncut"])
    t = dump_tree(bsp)
    rng = np.random.default_rng(21)
    lo, hi = t["lo"], t["hi"]
    O, D = [], []
    for i in range(len(lo)):                      # every node: origins on each of its six bound planes
        for ax in range(3):
            for bound in (lo[i, ax], hi[i, ax]):
                for _ in range(3):
                    o = rng.uniform(lo[0] - 1.0, hi[0] + 1.0)
                    o[ax] = bound
                    d = rng.normal(size=3)
                    d[ax] = 0.0                   # stays exactly zero under the normalisation
                    if rng.uniform() < 0.3:
                        d[(ax + 1) % 3] = 0.0
                    d /= np.linalg.norm(d)
                    O.append(o); D.append(d)
    O, D = np.array(O), np.array(D)
    with q4_mode(True):
        pts, ids = bsp.intersect_vectorize(ray_sources=O, ray_directions=D)
    save(out, "secondary_oncut", dict(n=int(len(O)), hits=int((ids != 0).sum())), origins=O, directions=D, points=pts,
         ids=ids.astype(np.int32))
    for w, h in ((65, 39), (9, 5)):
        src, dirs = ref_rf.get_initial_rays(cam, image_width=w, image_height=h)
        with q4_mode(False) as q4:
            pts, ids = bsp.intersect_vectorize(ray_sources=src, ray_directions=dirs)
        with q4_mode(True):
            pts_f, ids_f = bsp.intersect_vectorize(ray_sources=src, ray_directions=dirs)
        save(out, "primary_oncut_%dx%d" % (w, h),
             dict(width=w, height=h, q4_events=q4.events, ids_differ_q4=int((ids != ids_f).sum()),
                  zero_dx=int((dirs[:, 0] == 0).sum())),
             ids=ids.astype(np.int16), ids_q4fix=ids_f.astype(np.int16), points_q4fix=pts_f)
    g_frames(out, scenes, None, [("oncut_65x39_s1", "oncut", 65, 39, 1, None), ("oncut_65x39_s3", "oncut", 65, 39, 3, None),
                                 ("oncut_43x25_s2_r1", "oncut", 43, 25, 2, 1)])


def g_stochastic(out, scenes, K=32):
    """Mean of K seeded reference renders at the same s."""
    for name, tag, w, h, sh, rec in (("pool_192x108_s5", "pool", 192, 108, 5, None),
                                     ("pool_192x108_s2", "pool", 192, 108, 2, None)):
        acc = np.zeros((w * h, 3))
        acc8 = np.zeros((h, w, 3))
        t0 = time.time()
        for k in range(K):
            img, meta = render(scenes[tag], w, h, sh=sh, rec=rec, jitter=True, seed=1000 + k, fix_q4=True)
            acc += img.clip(0, 1)
            acc8 += quantize(img.clip(0, 1), h, w)
            if k == 0:
                first = quantize(img.clip(0, 1), h, w)
        meta.update(scene=tag, K=K, seconds=round(time.time() - t0, 1))
        save(out, "stoch_%s_mean%d" % (name, K), meta, mean_rgb=(acc / K).astype(np.float32).reshape(h, w, 3),
             mean_rgb8=(acc8 / K).astype(np.float32), first_rgb8=first)


def g_counts(out, scenes):
    res = {}
    for name, tag, w, h, sh, rec in (("pool_192x108_s1", "pool", 192, 108, 1, None),
                                     ("pool_100x100_s5", "pool", 100, 100, None, None),
                                     ("original_60x60", "original", 60, 60, None, None),
                                     ("small_80x50", "small", 80, 50, None, None)):
        _, meta = render(scenes[tag], w, h, sh=sh, rec=rec, jitter=False, fix_q4=True)
        res[name] = meta
        print("  counts", name, meta, flush=True)
    # numbers measured by the survey session for the full-size configs
    res["pool_1920x1080_s1"] = dict(width=1920, height=1080, sh_rays=1, rec_max=10, rays=17380382, intersect_calls=66,
                                    source="BASELINE.md")
    res["original_500x500"] = dict(width=500, height=500, sh_rays=3, rec_max=10, rays=7325032, intersect_calls=2512,
                                   source="BASELINE.md")
    with open(os.path.join(out, "counts.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="")
    ap.add_argument("--frames", default="")
    ap.add_argument("-K", type=int, default=32)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    sc_dir = os.path.join(a.out, "scenes")
    os.makedirs(sc_dir, exist_ok=True)
    # scene fixtures: the two data files of the reference + generated ones
    scenes = {}
    for tag in ("pool", "original"):
        dst = os.path.join(sc_dir, tag + ".txt")
        if not os.path.exists(dst):
            with open(os.path.join(REF, "scenes", tag + ".txt")) as f, open(dst, "w") as g:
                g.write(f.read())
        scenes[tag] = dst
    gen = {"small": lambda: gen_scene.small_mixed(7, 200),
           "synth400": lambda: gen_scene.synth_spheres(20, 1234, 2, 3),
           "oncut": gen_scene.on_cut_symmetric}
    for tag, fn in gen.items():
        dst = os.path.join(sc_dir, tag + ".txt")
        if not os.path.exists(dst):
            with open(dst, "w") as g:
                g.write(fn())
        scenes[tag] = dst

    steps = dict(rays=g_rays, prims=g_prims, bsp=g_bsp, shadow_grid=g_shadow_grid, shade_terms=g_shade_terms,
                 secondary=g_secondary, visibility=g_visibility, primary=g_primary, counts=g_counts,
                 frames=lambda o, s: g_frames(o, s, set(a.frames.split(",")) if a.frames else None),
                 stochastic=lambda o, s: g_stochastic(o, s, a.K), oncut=g_oncut)
    only = [x for x in a.only.split(",") if x]
    for name, fn in steps.items():
        if only and name not in only:
            continue
        t0 = time.time()
        print("[%s]" % name, flush=True)
        fn(a.out, scenes)
        print("[%s] done in %.1f s" % (name, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
