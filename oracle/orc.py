"""ctypes wrapper of oracle/liboracle.so - TEST INFRASTRUCTURE ONLY (see the
header of oracle.cpp).  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

import ray_tracing_amd  # noqa: F401  (struct definitions + host mirror on sys.path)
from ray_tracing_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
COUNTER_NAMES = ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits", "node_tests",
                 "sphere_tests", "box_tests", "plane_tests", "shade_lights", "shadow_node_tests",
                 "shadow_sphere_tests", "shadow_box_tests", "shadow_plane_tests")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_tile_layout.restype = C.c_int64
    return _LIB


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _counters(raw):
    d = dict(zip(COUNTER_NAMES, (int(x) for x in raw)))
    d["rays"] = d["primary_rays"] + d["shadow_rays"] + d["reflect_rays"] + d["transmit_rays"]
    return d


def intersect(packed, origins, directions, threads=0):
    o, d = _f64(origins), _f64(directions)
    n = o.shape[0]
    pts = np.empty((n, 3))
    ids = np.empty(n, dtype=np.int64)
    cnt = np.zeros(14, dtype=np.int64)
    lib().orc_intersect(C.byref(packed.desc), _p(o), _p(d), C.c_int64(n), _p(pts), _p(ids, C.c_int64),
                        _p(cnt, C.c_int64), C.c_int(threads))
    return pts, ids, _counters(cnt)


def render(packed, params, threads=0, want_f64=True):
    """-> dict(rgb (npix,3) float64 unclipped, rgb8 (npix,3) uint8, ids (npix,), counters)."""
    npix = int(lib().orc_tile_layout(C.byref(params), None))
    rgb = np.empty((npix, 3)) if want_f64 else None
    rgb8 = np.empty((npix, 3), dtype=np.uint8)
    ids = np.empty(npix, dtype=np.int32)
    cnt = np.zeros(14, dtype=np.int64)
    lib().orc_render(C.byref(packed.desc), C.byref(params), _p(rgb) if want_f64 else None, _p(rgb8, C.c_uint8),
                     _p(ids, C.c_int32), _p(cnt, C.c_int64), C.c_int(threads))
    return dict(rgb=rgb, rgb8=rgb8, ids=ids, counters=_counters(cnt))


def trace(packed, params, origins, directions, threads=0):
    o, d = _f64(origins), _f64(directions)
    n = o.shape[0]
    rgb = np.empty((n, 3))
    cnt = np.zeros(14, dtype=np.int64)
    lib().orc_trace(C.byref(packed.desc), C.byref(params), _p(o), _p(d), C.c_int64(n), _p(rgb), _p(cnt, C.c_int64),
                    C.c_int(threads))
    return rgb, _counters(cnt)


def tile_layout(params):
    n = int(lib().orc_tile_layout(C.byref(params), None))
    pix = np.empty(n, dtype=np.int64)
    lib().orc_tile_layout(C.byref(params), _p(pix, C.c_int64))
    return pix


def initial_rays(packed, width, height):
    src = np.empty((width * height, 3))
    dirs = np.empty((width * height, 3))
    lib().orc_initial_rays(C.byref(packed.desc), C.c_int(width), C.c_int(height), _p(src), _p(dirs))
    return src, dirs


def sphere_hits(o, d, c, r2):
    o, d = _f64(o), _f64(d)
    out = np.empty_like(o)
    lib().orc_sphere_hits(_p(o), _p(d), C.c_int64(len(o)), _p(_f64(c)), C.c_double(r2), _p(out))
    return out


def box_hits(o, d, lo, hi):
    o, d = _f64(o), _f64(d)
    out = np.empty_like(o)
    lib().orc_box_hits(_p(o), _p(d), C.c_int64(len(o)), _p(_f64(lo)), _p(_f64(hi)), _p(out))
    return out


def plane_hits(o, d, n, dd):
    o, d = _f64(o), _f64(d)
    out = np.empty_like(o)
    lib().orc_plane_hits(_p(o), _p(d), C.c_int64(len(o)), _p(_f64(n)), C.c_double(dd), _p(out))
    return out


def slab_mask(o, d, lo, hi, outside_only):
    o, d = _f64(o), _f64(d)
    mask = np.empty(len(o), dtype=np.uint8)
    tn = np.empty(len(o))
    lib().orc_slab_mask(_p(o), _p(d), C.c_int64(len(o)), _p(_f64(lo)), _p(_f64(hi)), C.c_int(1 if outside_only else 0),
                        _p(mask, C.c_uint8), _p(tn))
    return mask.astype(bool), tn


def box_normals(pts, c, scale):
    pts = _f64(pts)
    out = np.empty_like(pts)
    lib().orc_box_normals(_p(pts), C.c_int64(len(pts)), _p(_f64(c)), C.c_double(scale), _p(out))
    return out


def sphere_normals(pts, c):
    pts = _f64(pts)
    out = np.empty_like(pts)
    lib().orc_sphere_normals(_p(pts), C.c_int64(len(pts)), _p(_f64(c)), _p(out))
    return out


def shadow_grid(hits, light_pos, radius, s):
    hits = _f64(hits)
    n = len(hits)
    src = np.empty((n, s, s, 3))
    dirs = np.empty((n, s, s, 3))
    lib().orc_shadow_grid(_p(hits), C.c_int64(n), _p(_f64(light_pos)), C.c_double(radius), C.c_int(s), _p(src),
                          _p(dirs))
    return src, dirs


def shade_terms(N, L, V, Kd, Ks, Imd, Ims, alpha):
    N, L, V, Kd, Ks, Imd, Ims = (_f64(x) for x in (N, L, V, Kd, Ks, Imd, Ims))
    alpha = np.ascontiguousarray(alpha, dtype=np.int64)
    n, Lc = N.shape[0], L.shape[0]
    diff = np.empty((n, 3))
    spec = np.empty((n, 3))
    lib().orc_shade_terms(C.c_int64(n), C.c_int(Lc), _p(N), _p(L), _p(V), _p(Kd), _p(Ks), _p(Imd), _p(Ims),
                          _p(alpha, C.c_int64), _p(diff), _p(spec))
    return diff, spec


def jitter6(seed, pixel, path, light, sample, g):
    out = np.empty(6)
    lib().orc_jitter6(C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(path), C.c_uint32(light), C.c_uint32(sample),
                      C.c_double(g), _p(out))
    return out
