"""Flatten the host scene objects into the SoA float64 buffers of rt_scene_desc
(include/rt_amd.h).  Duck-typed on the attributes the reference's classes carry
(SURVEY.md §8b), so objects of the reference's own classes pack as well.

Everything the reference derives once per object on the host is derived here
with the very same NumPy / Python expressions, so the device sees bit-identical
constants: sphere `radius ** 2` (Sphere.py:79), box bounds `position -/+ scale/2`
(Cube.py:89-90), integer shininess (Material.py:49,58), the stored plane normal
and offset (InfinitePlane.py:8-10) and the camera basis (Camera.py:13).
"""
import ctypes as C

import numpy as np

from . import abi


def _kind_of(obj):
    if hasattr(obj, "radius") and hasattr(obj, "position"):
        return abi.RT_SURF_SPHERE
    if hasattr(obj, "scale") and hasattr(obj, "position"):
        return abi.RT_SURF_BOX
    if hasattr(obj, "normal") and hasattr(obj, "offset"):
        return abi.RT_SURF_PLANE
    return abi.RT_SURF_BACKGROUND


def _is_finite(obj):
    return hasattr(obj, "get_enclosing_values") and _kind_of(obj) in (abi.RT_SURF_SPHERE, abi.RT_SURF_BOX)


class PackedScene:
    """Owns the NumPy buffers a SceneDesc points into."""

    def __init__(self):
        self.arrays = {}
        self.desc = abi.SceneDesc()
        self.desc.abi_version = abi.RT_AMD_ABI_VERSION

    def _set(self, name, arr, dtype):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        self.arrays[name] = arr
        ctype = C.c_double if dtype == np.float64 else C.c_int32
        setattr(self.desc, name, arr.ctypes.data_as(C.POINTER(ctype)))
        return arr

    def f64(self, name, arr, cols=None):
        arr = np.asarray(arr, dtype=np.float64)
        if cols:
            arr = arr.reshape(-1, cols)
        return self._set(name, arr, np.float64)

    def i32(self, name, arr):
        return self._set(name, np.asarray(arr, dtype=np.int64), np.int32)

    @property
    def counts(self):
        d = self.desc
        return dict(surfaces=d.n_surfaces, spheres=d.n_spheres, boxes=d.n_boxes, planes=d.n_planes,
                    materials=d.n_materials, lights=d.n_lights, nodes=d.n_nodes, leaf_refs=d.n_leaf_refs)


def pack_scene(surfaces, materials=None, lights=None, camera=None, settings=None, bsp_tree=None):
    """-> PackedScene.  `bsp_tree` is the root returned by host BSPNode.build_bsp_tree
    (built here when None).  materials/lights/camera/settings may be None for
    intersect-only contexts."""
    ps = PackedScene()
    d = ps.desc
    surfaces = list(surfaces)
    if bsp_tree is None:
        import importlib
        bsp_tree = importlib.import_module("BSPNode").BSPNode.build_bsp_tree(surfaces)
    flat = bsp_tree.flat
    finite = bsp_tree.finite_objects

    # --- geometry tables -------------------------------------------------------------------
    spheres = [o for o in surfaces if _kind_of(o) == abi.RT_SURF_SPHERE]
    boxes = [o for o in surfaces if _kind_of(o) == abi.RT_SURF_BOX]
    # order of the root node's plane list (BSPNode.py:111), Background excluded
    root_planes = [o for o in (bsp_tree.inf_plains or []) if _kind_of(o) == abi.RT_SURF_PLANE]
    plane_row = {id(o): k for k, o in enumerate(root_planes)}
    all_planes = root_planes
    sph_row = {id(o): k for k, o in enumerate(spheres)}
    box_row = {id(o): k for k, o in enumerate(boxes)}

    d.n_spheres = len(spheres)
    ps.f64("sph_center", [np.asarray(o.position, dtype=np.float64) for o in spheres] or np.zeros((0, 3)), 3)
    ps.f64("sph_radius", [float(o.radius) for o in spheres])
    ps.f64("sph_r2", [o.radius ** 2 for o in spheres])
    ps.i32("sph_id", [o.index for o in spheres])

    d.n_boxes = len(boxes)
    ps.f64("box_center", [np.asarray(o.position, dtype=np.float64) for o in boxes] or np.zeros((0, 3)), 3)
    ps.f64("box_scale", [float(o.scale) for o in boxes])
    ps.f64("box_lo", [np.asarray(o.position - o.scale / 2, dtype=np.float64) for o in boxes] or np.zeros((0, 3)), 3)
    ps.f64("box_hi", [np.asarray(o.position + o.scale / 2, dtype=np.float64) for o in boxes] or np.zeros((0, 3)), 3)
    ps.i32("box_id", [o.index for o in boxes])

    d.n_planes = len(root_planes)
    ps.f64("pln_normal", [np.asarray(o.normal, dtype=np.float64) for o in all_planes] or np.zeros((0, 3)), 3)
    ps.f64("pln_d", [float(o.offset) for o in all_planes])
    ps.i32("pln_id", [o.index for o in all_planes])
    d.far_clip = 1 if any(_kind_of(o) == abi.RT_SURF_BACKGROUND for o in (bsp_tree.inf_plains or [])) else 0

    # --- surfaces by list position (normals / materials are looked up by position, Q8) -------
    d.n_surfaces = len(surfaces)
    kinds, geoms, mats = [], [], []
    for o in surfaces:
        k = _kind_of(o)
        kinds.append(k)
        geoms.append({abi.RT_SURF_SPHERE: sph_row, abi.RT_SURF_BOX: box_row, abi.RT_SURF_PLANE: plane_row}
                     .get(k, {}).get(id(o), -1))
        mats.append(int(o.get_material_index()) if hasattr(o, "get_material_index") else int(o.material_index or 0))
    ps.i32("surf_kind", kinds)
    ps.i32("surf_geom", geoms)
    ps.i32("surf_mat", mats)

    # --- materials -------------------------------------------------------------------------
    materials = list(materials or [])
    d.n_materials = len(materials)
    ps.f64("mat_kd", [np.asarray(m.diffuse_color, dtype=np.float64) for m in materials] or np.zeros((0, 3)), 3)
    ps.f64("mat_ks", [np.asarray(m.specular_color, dtype=np.float64) for m in materials] or np.zeros((0, 3)), 3)
    ps.f64("mat_kr", [np.asarray(m.reflection_color, dtype=np.float64) for m in materials] or np.zeros((0, 3)), 3)
    ps.i32("mat_phong", [int(float(m.shininess)) for m in materials])       # C-cast truncation toward zero
    ps.f64("mat_transparency", [float(m.transparency) for m in materials])

    # --- lights ----------------------------------------------------------------------------
    lights = list(lights or [])
    d.n_lights = len(lights)
    ps.f64("lgt_pos", [np.asarray(l.position, dtype=np.float64) for l in lights] or np.zeros((0, 3)), 3)
    ps.f64("lgt_color", [np.asarray(l.color, dtype=np.float64) for l in lights] or np.zeros((0, 3)), 3)
    ps.f64("lgt_spec", [float(l.specular_intensity) for l in lights])
    ps.f64("lgt_shadow", [float(l.shadow_intensity) for l in lights])
    ps.f64("lgt_radius", [float(l.radius) for l in lights])

    # --- tree ------------------------------------------------------------------------------
    prim_of_finite = np.zeros(len(finite), dtype=np.int64)
    for pos, o in enumerate(finite):
        prim_of_finite[pos] = sph_row[id(o)] if _kind_of(o) == abi.RT_SURF_SPHERE else len(spheres) + box_row[id(o)]
    d.n_nodes = len(flat.right)
    ps.f64("node_lo", flat.lo, 3)
    ps.f64("node_hi", flat.hi, 3)
    ps.i32("node_right", flat.right)
    ps.i32("node_first", flat.first)
    ps.i32("node_count", flat.count)
    d.n_leaf_refs = len(flat.leaf_prim)
    ps.i32("leaf_prim", prim_of_finite[flat.leaf_prim] if len(flat.leaf_prim) else [])

    # --- camera / settings -----------------------------------------------------------------
    if camera is not None:
        for name, vec in (("cam_pos", camera.position), ("cam_x", camera.x_dir), ("cam_y", camera.y_dir),
                          ("cam_z", camera.z_dir)):
            setattr(d, name, (C.c_double * 3)(*[float(x) for x in vec]))
        d.cam_screen_distance = float(camera.screen_distance)
        d.cam_screen_width = float(camera.screen_width)
    if settings is not None:
        d.background = (C.c_double * 3)(*[float(x) for x in settings.background_color])
        d.sh_rays = int(settings.root_number_shadow_rays)
        d.rec_max = int(settings.max_recursions)
    return ps
