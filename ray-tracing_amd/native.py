"""ctypes binding of csrc/librt_amd.so (the C-ABI of include/rt_amd.h).

There is no CPU fallback: if the library is missing or no GPU is visible every
call raises.  `python __graft_entry__.py` (or `make -C ray-tracing_amd/csrc`)
builds the library in-tree.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_PKG = os.path.dirname(os.path.abspath(abi.__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_PKG, "csrc", "librt_amd.so")    # RT_AMD_LIB: A/B runs of two builds (tools/)
_LIB = None

EXPORTS = ("rt_abi_version", "rt_create", "rt_destroy", "rt_last_error", "rt_tile_pixels", "rt_tile_layout",
           "rt_render", "rt_render_mean", "rt_render_device", "rt_render_device_noalloc", "rt_noalloc_scratch_bytes",
           "rt_debug_alloc_calls", "rt_debug_tile_frames", "rt_debug_early_frames", "rt_sync_stats", "rt_voided_frames", "rt_frame_bytes", "rt_trace", "rt_intersect", "rt_visibility", "rt_intersect_device", "rt_visibility_device",
           "rt_deinterleave_u8",
           "rt_initial_rays", "rt_unit_prim_intersect", "rt_unit_prim_normals", "rt_unit_slab_mask",
           "rt_bsp_build", "rt_bsp_stats_get", "rt_bsp_fetch", "rt_bsp_destroy", "rt_bsp_last_error",
           "rt_scene_parse", "rt_scene_parse_file", "rt_scene_get_desc", "rt_scene_get_info", "rt_scene_corners",
           "rt_scene_light_ids", "rt_scene_set_tree", "rt_scene_build_tree", "rt_scene_free", "rt_scene_last_error")


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (abi.STATUS_NAMES.get(code, "RT_ERR"), code, message))
        self.code = code


def lib():
    """Load the HIP library; raises (never falls back) when it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing - build it with `python __graft_entry__.py` or "
                               "`make -C ray-tracing_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_last_error.argtypes = [C.c_void_p]
        L.rt_tile_pixels.restype = C.c_int64
        L.rt_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.rt_destroy.argtypes = [C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(abi.Stats)]
        L.rt_render_mean.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_int32, C.c_void_p, C.c_void_p,
                                     C.POINTER(abi.Stats)]
        L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.rt_render_device_noalloc.argtypes = L.rt_render_device.argtypes
        L.rt_noalloc_scratch_bytes.restype = C.c_int64
        L.rt_debug_alloc_calls.restype = C.c_int64
        L.rt_debug_tile_frames.restype = C.c_int64
        L.rt_debug_tile_frames.argtypes = [C.c_void_p]
        L.rt_debug_early_frames.restype = C.c_int64
        L.rt_debug_early_frames.argtypes = [C.c_void_p]
        L.rt_sync_stats.argtypes = [C.c_void_p, C.POINTER(abi.Stats)]
        L.rt_frame_bytes.restype = C.c_int64
        L.rt_frame_bytes.argtypes = [C.c_void_p]
        L.rt_voided_frames.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.rt_trace.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p, C.POINTER(abi.Stats)]
        L.rt_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.rt_visibility.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.rt_intersect_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_visibility_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.rt_deinterleave_u8.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p]
        L.rt_initial_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rt_tile_layout.argtypes = [C.POINTER(abi.RenderParams), C.c_void_p]
        L.rt_tile_pixels.argtypes = [C.POINTER(abi.RenderParams)]
        L.rt_unit_prim_intersect.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_void_p]
        L.rt_unit_prim_normals.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.rt_unit_slab_mask.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p]
        L.rt_bsp_last_error.restype = C.c_char_p
        L.rt_bsp_last_error.argtypes = [C.c_void_p]
        L.rt_bsp_build.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_double, C.POINTER(C.c_void_p)]
        L.rt_bsp_stats_get.argtypes = [C.c_void_p, C.POINTER(abi.BspStats)]
        L.rt_bsp_fetch.argtypes = [C.c_void_p] * 7
        L.rt_bsp_destroy.argtypes = [C.c_void_p]
        L.rt_scene_last_error.restype = C.c_char_p
        L.rt_scene_parse.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_parse_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.rt_scene_get_desc.restype = C.POINTER(abi.SceneDesc)
        L.rt_scene_get_desc.argtypes = [C.c_void_p]
        L.rt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(abi.SceneInfo)]
        L.rt_scene_corners.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)),
                                       C.POINTER(C.c_int64)]
        L.rt_scene_light_ids.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]
        L.rt_scene_set_tree.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p]
        L.rt_scene_build_tree.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.BspStats)]
        L.rt_scene_free.argtypes = [C.c_void_p]
        L.rt_scene_free.restype = None
        if L.rt_abi_version() != abi.RT_AMD_ABI_VERSION:
            raise RuntimeError("librt_amd.so ABI version mismatch")
        _LIB = L
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _check(code, ctx=None):
    if code != abi.RT_OK:
        msg = lib().rt_last_error(ctx)
        raise NativeError(code, msg.decode() if msg else "")


def tile_pixels(params):
    n = lib().rt_tile_pixels(C.byref(params))
    if n < 0:
        raise NativeError(-1, "bad render params")
    return int(n)


def tile_layout(params):
    pix = np.empty(tile_pixels(params), dtype=np.int64)
    _check(lib().rt_tile_layout(C.byref(params), pix.ctypes.data))
    return pix


class Context:
    """A scene resident on one GPU (rt_create / rt_destroy)."""

    def __init__(self, packed, device=0):
        self._h = C.c_void_p()
        self.packed = packed        # keeps the host arrays alive
        self.device = device
        _check(lib().rt_create(C.byref(packed.desc), int(device), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- whole frames ---------------------------------------------------------------------
    def render(self, params, want_rgb=True, want_ids=True):
        """Blocking render into host arrays: dict(rgb float32 (n,3), rgb8 uint8 (n,3), ids int32 (n,), stats)."""
        n = tile_pixels(params)
        rgb = np.empty((n, 3), dtype=np.float32) if want_rgb else None
        rgb8 = np.empty((n, 3), dtype=np.uint8)
        ids = np.empty(n, dtype=np.int32) if want_ids else None
        st = abi.Stats()
        _check(lib().rt_render(self._h, C.byref(params), rgb.ctypes.data if want_rgb else None, rgb8.ctypes.data,
                               ids.ctypes.data if want_ids else None, C.byref(st)), self._h)
        return dict(rgb=rgb, rgb8=rgb8, ids=ids, stats=st.as_dict())

    def render_mean(self, params, frames, want_rgb=True):
        """Mean of the clipped frames for seed .. seed+frames-1, accumulated on the device (rt_render_mean):
        dict(rgb float32 (n,3), rgb8 uint8 (n,3), stats summed over the frames)."""
        n = tile_pixels(params)
        rgb = np.empty((n, 3), dtype=np.float32) if want_rgb else None
        rgb8 = np.empty((n, 3), dtype=np.uint8)
        st = abi.Stats()
        _check(lib().rt_render_mean(self._h, C.byref(params), int(frames), rgb.ctypes.data if want_rgb else None,
                                    rgb8.ctypes.data, C.byref(st)), self._h)
        return dict(rgb=rgb, rgb8=rgb8, stats=st.as_dict())

    def render_device(self, params, d_rgb=0, d_rgb8=0, d_ids=0, stream=0):
        """Enqueue a frame; outputs are raw device pointers (ints, 0 = not wanted); returns immediately."""
        _check(lib().rt_render_device(self._h, C.byref(params), C.c_void_p(d_rgb or None), C.c_void_p(d_rgb8 or None),
                                      C.c_void_p(d_ids or None), C.c_void_p(stream or None)), self._h)

    def render_device_noalloc(self, params, d_rgb=0, d_rgb8=0, d_ids=0, stream=0):
        """rt_render_device_noalloc: the same frame with all of its working memory in the library's static device array."""
        _check(lib().rt_render_device_noalloc(self._h, C.byref(params), C.c_void_p(d_rgb or None), C.c_void_p(d_rgb8 or None),
                                              C.c_void_p(d_ids or None), C.c_void_p(stream or None)), self._h)

    def sync_stats(self):
        st = abi.Stats()
        _check(lib().rt_sync_stats(self._h, C.byref(st)), self._h)
        return st.as_dict()

    def frame_bytes(self):
        return int(lib().rt_frame_bytes(self._h))

    def voided_frames(self, reset=False):
        """Frames voided by a queue overflow since creation / the last reset (device counter; synchronises)."""
        n = C.c_int64()
        _check(lib().rt_voided_frames(self._h, 1 if reset else 0, C.byref(n)), self._h)
        return int(n.value)

    def deinterleave_u8(self, params, d_packed, stride_bytes, d_image, stream=0):
        _check(lib().rt_deinterleave_u8(self._h, C.byref(params), C.c_void_p(d_packed), int(stride_bytes),
                                        C.c_void_p(d_image), C.c_void_p(stream or None)), self._h)

    # -- ray batches ----------------------------------------------------------------------
    def trace(self, params, origins, directions):
        """ray_tracing() for an arbitrary batch -> ((N,3) float64 unclipped colours, stats)."""
        o, d = _f64(origins), _f64(directions)
        if o.shape != d.shape or o.ndim != 2 or o.shape[1] != 3:
            raise ValueError("origins/directions must both be (N,3)")
        out = np.empty_like(o)
        st = abi.Stats()
        _check(lib().rt_trace(self._h, C.byref(params), o.ctypes.data, d.ctypes.data, o.shape[0], out.ctypes.data,
                              C.byref(st)), self._h)
        return out, st.as_dict()

    def intersect(self, origins, directions):
        """BSPNode.intersect_vectorize -> ((N,3) float64 points, (N,) int64 ids, 0 = miss)."""
        o, d = _f64(origins), _f64(directions)
        if o.shape != d.shape or o.ndim != 2 or o.shape[1] != 3:
            raise ValueError("ray_sources/ray_directions must both be (N,3)")
        pts = np.empty_like(o)
        ids = np.empty(o.shape[0], dtype=np.int64)
        _check(lib().rt_intersect(self._h, o.ctypes.data, d.ctypes.data, o.shape[0], pts.ctypes.data,
                                  ids.ctypes.data), self._h)
        return pts, ids

    def visibility(self, origins, directions, targets):
        """The lit test of Light.compute_shadows (Light.py:214-221) for a ray batch: (N,) bool,
        True where the nearest hit of ray i lies within 1e-6 of targets[i]."""
        o, d, t = _f64(origins), _f64(directions), _f64(targets)
        if not (o.shape == d.shape == t.shape) or o.ndim != 2 or o.shape[1] != 3:
            raise ValueError("ray_sources/ray_directions/targets must all be (N,3)")
        lit = np.empty(o.shape[0], dtype=np.uint8)
        _check(lib().rt_visibility(self._h, o.ctypes.data, d.ctypes.data, t.ctypes.data, o.shape[0],
                                   lit.ctypes.data), self._h)
        return lit.astype(bool)

    def intersect_device(self, d_origins, d_directions, n, d_points, d_ids, stream=0):
        """rt_intersect_device: raw device pointers (ints), float64 [n*3] in / out, int64 [n] ids; enqueued only."""
        _check(lib().rt_intersect_device(self._h, C.c_void_p(d_origins), C.c_void_p(d_directions), int(n), C.c_void_p(d_points),
                                         C.c_void_p(d_ids), C.c_void_p(stream or None)), self._h)

    def visibility_device(self, d_origins, d_directions, d_targets, n, d_lit, stream=0):
        """rt_visibility_device: raw device pointers (ints); uint8 [n] verdicts; enqueued only."""
        _check(lib().rt_visibility_device(self._h, C.c_void_p(d_origins), C.c_void_p(d_directions), C.c_void_p(d_targets), int(n),
                                          C.c_void_p(d_lit), C.c_void_p(stream or None)), self._h)

    def initial_rays(self, width, height):
        src = np.empty((width * height, 3))
        dirs = np.empty((width * height, 3))
        _check(lib().rt_initial_rays(self._h, int(width), int(height), src.ctypes.data, dirs.ctypes.data), self._h)
        return src, dirs


# -- single-primitive kernels (host mirror's surface.intersect_vectorized etc.) ----------------
def prim_intersect(kind, params12, origins, directions, device=0):
    o, d, par = _f64(origins), _f64(directions), _f64(params12)
    out = np.empty_like(o)
    _check(lib().rt_unit_prim_intersect(device, int(kind), par.ctypes.data, o.ctypes.data, d.ctypes.data, o.shape[0],
                                        out.ctypes.data))
    return out


def prim_normals(kind, params12, points, device=0):
    p, par = _f64(points), _f64(params12)
    out = np.empty_like(p)
    _check(lib().rt_unit_prim_normals(device, int(kind), par.ctypes.data, p.ctypes.data, p.shape[0],
                                      out.ctypes.data))
    return out


def slab_mask(origins, directions, bounds_min, bounds_max, outside_hits_only=True, device=0):
    o, d = _f64(origins), _f64(directions)
    lo, hi = _f64(bounds_min), _f64(bounds_max)
    mask = np.empty(o.shape[0], dtype=np.uint8)
    tn = np.empty(o.shape[0])
    _check(lib().rt_unit_slab_mask(device, lo.ctypes.data, hi.ctypes.data, 1 if outside_hits_only else 0,
                                   o.ctypes.data, d.ctypes.data, o.shape[0], mask.ctypes.data, tn.ctypes.data))
    return mask.astype(bool)[:, None], tn[:, None]


# -- tree build on the GPU (BSPNode.build_bsp_tree, BSPNode.py:54-121) ---------------------------
def bsp_build(small_corner, big_corner, root_lo=None, root_hi=None, leaf_size=10, epsilon=1e-6, device=0):
    """-> dict(lo (n,3), hi (n,3), right, first, count, leaf_prim (int32), stats).  The tree is built in HBM
    by rt_bsp_build and copied back in the DFS pre-order layout of rt_scene_desc; leaf_prim holds
    positions into the corner arrays."""
    small, big = _f64(small_corner).reshape(-1, 3), _f64(big_corner).reshape(-1, 3)
    if small.shape != big.shape:
        raise ValueError("small_corner/big_corner must both be (N,3)")
    rlo = None if root_lo is None else _f64(root_lo).reshape(3)
    rhi = None if root_hi is None else _f64(root_hi).reshape(3)
    h = C.c_void_p()
    rc = lib().rt_bsp_build(int(device), small.shape[0], small.ctypes.data, big.ctypes.data,
                            None if rlo is None else rlo.ctypes.data, None if rhi is None else rhi.ctypes.data,
                            int(leaf_size), float(epsilon), C.byref(h))
    if rc != abi.RT_OK:
        msg = lib().rt_bsp_last_error(None)
        raise NativeError(rc, msg.decode() if msg else "")
    try:
        st = abi.BspStats()
        lib().rt_bsp_stats_get(h, C.byref(st))
        n, r = st.n_nodes, st.n_leaf_refs
        out = dict(lo=np.empty((n, 3)), hi=np.empty((n, 3)), right=np.empty(n, dtype=np.int32),
                   first=np.empty(n, dtype=np.int32), count=np.empty(n, dtype=np.int32),
                   leaf_prim=np.empty(r, dtype=np.int32))
        rc = lib().rt_bsp_fetch(h, *[out[k].ctypes.data for k in ("lo", "hi", "right", "first", "count", "leaf_prim")])
        if rc != abi.RT_OK:
            msg = lib().rt_bsp_last_error(h)
            raise NativeError(rc, msg.decode() if msg else "")
        out["stats"] = st.as_dict()
        return out
    finally:
        lib().rt_bsp_destroy(h)


# -- scene ingestion without Python objects (Parser.parse_scene_file, Parser.py:25-71) ------------
class NativeScene:
    """Scene tables made by rt_scene_parse / rt_scene_parse_file.  Quacks like scene_pack.PackedScene
    (`desc`, `arrays`, `counts`), so native.Context(scene) and the oracle wrapper take it as is.  The tree is
    attached by build_tree (GPU) or set_tree (a FlatTree of the host builder)."""

    def __init__(self, text=None, path=None, threads=0):
        self._h = C.c_void_p()
        if (text is None) == (path is None):
            raise ValueError("give either text or path")
        if path is not None:
            rc = lib().rt_scene_parse_file(os.fsencode(path), int(threads), C.byref(self._h))
        else:
            raw = text.encode() if isinstance(text, str) else bytes(text)
            rc = lib().rt_scene_parse(raw, len(raw), int(threads), C.byref(self._h))
        if rc != abi.RT_OK:
            msg = (lib().rt_scene_last_error() or b"").decode()
            if rc == -1:
                raise ValueError(msg)           # the reference's parser raises ValueError for bad lines
            raise NativeError(rc, msg)
        self.desc = lib().rt_scene_get_desc(self._h).contents      # lives inside the native object
        info = abi.SceneInfo()
        lib().rt_scene_get_info(self._h, C.byref(info))
        self.info = info.as_dict()
        self.build_stats = None

    def close(self):
        if self._h:
            lib().rt_scene_free(self._h)
            self._h = C.c_void_p()
            self.desc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _view(self, ptr, rows, cols):
        if rows == 0:
            return np.zeros((0, cols) if cols > 1 else (0,), dtype=np.float64 if ptr._type_ is C.c_double else np.int32)
        a = np.ctypeslib.as_array(ptr, shape=(rows * cols,))
        return a.reshape(rows, cols) if cols > 1 else a

    @property
    def arrays(self):
        """name -> NumPy view of the table (valid until close(); re-read after attaching a tree)."""
        return {name: self._view(getattr(self.desc, name), getattr(self.desc, cnt), cols)
                for name, (cnt, cols) in abi.SCENE_TABLES.items()}

    @property
    def counts(self):
        d = self.desc
        return dict(surfaces=d.n_surfaces, spheres=d.n_spheres, boxes=d.n_boxes, planes=d.n_planes,
                    materials=d.n_materials, lights=d.n_lights, nodes=d.n_nodes, leaf_refs=d.n_leaf_refs)

    def corners(self):
        """Object3D.get_enclosing_values of the finite objects in list order -> (small (n,3), big (n,3)) views."""
        ps, pb, n = C.POINTER(C.c_double)(), C.POINTER(C.c_double)(), C.c_int64()
        lib().rt_scene_corners(self._h, C.byref(ps), C.byref(pb), C.byref(n))
        return self._view(ps, n.value, 3), self._view(pb, n.value, 3)

    def ids(self):
        """(Light.index per light, Material.index per material row)."""
        pl, pm = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        lib().rt_scene_light_ids(self._h, C.byref(pl), C.byref(pm))
        return self._view(pl, self.desc.n_lights, 1).copy(), self._view(pm, self.desc.n_materials, 1).copy()

    def build_tree(self, device=0):
        st = abi.BspStats()
        rc = lib().rt_scene_build_tree(self._h, int(device), C.byref(st))
        if rc != abi.RT_OK:
            raise NativeError(rc, (lib().rt_scene_last_error() or b"").decode())
        self.build_stats = st.as_dict()
        return self.build_stats

    def set_tree(self, flat):
        """flat: FlatTree of BSPNode.build_bsp_tree (leaf_prim = positions in the finite-object order)."""
        lo, hi = _f64(flat.lo), _f64(flat.hi)
        right, first, count, leaf = (np.ascontiguousarray(a, dtype=np.int32)
                                     for a in (flat.right, flat.first, flat.count, flat.leaf_prim))
        rc = lib().rt_scene_set_tree(self._h, len(right), lo.ctypes.data, hi.ctypes.data, right.ctypes.data,
                                     first.ctypes.data, count.ctypes.data, len(leaf), leaf.ctypes.data)
        if rc != abi.RT_OK:
            raise NativeError(rc, (lib().rt_scene_last_error() or b"").decode())
