"""ctypes mirror of include/rt_amd.h (structs and constants only, no library
loading).  Kept field-for-field in the order of the header."""
import ctypes as C

RT_AMD_ABI_VERSION = 2
RT_USE_SCENE_VALUE = -2147483648
RT_SURF_BACKGROUND, RT_SURF_SPHERE, RT_SURF_BOX, RT_SURF_PLANE = 0, 1, 2, 3
RT_FLAG_EXACT_TRAVERSAL = 1
RT_FLAG_PROFILE = 2
RT_FLAG_NO_BEAM = 4
RT_FLAG_NO_OVERLAP = 8
RT_FLAG_NO_SMALL = 16
RT_FLAG_TIME_DOMINANT = 32
RT_FLAG_NO_TILES = 64

RT_OK = 0
RT_ERR_INVALID_ARG, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_OOM, RT_ERR_QUEUE_OVERFLOW, RT_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
STATUS_NAMES = {0: "RT_OK", -1: "RT_ERR_INVALID_ARG", -2: "RT_ERR_NO_DEVICE", -3: "RT_ERR_HIP", -4: "RT_ERR_OOM",
                -5: "RT_ERR_QUEUE_OVERFLOW", -6: "RT_ERR_UNSUPPORTED"}

_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)
_d3 = C.c_double * 3


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_surfaces", C.c_int32), ("surf_kind", _pi), ("surf_geom", _pi), ("surf_mat", _pi),
        ("n_spheres", C.c_int32), ("sph_center", _pd), ("sph_radius", _pd), ("sph_r2", _pd), ("sph_id", _pi),
        ("n_boxes", C.c_int32), ("box_center", _pd), ("box_scale", _pd), ("box_lo", _pd), ("box_hi", _pd),
        ("box_id", _pi),
        ("n_planes", C.c_int32), ("pln_normal", _pd), ("pln_d", _pd), ("pln_id", _pi), ("far_clip", C.c_int32),
        ("n_materials", C.c_int32), ("mat_kd", _pd), ("mat_ks", _pd), ("mat_kr", _pd), ("mat_phong", _pi),
        ("mat_transparency", _pd),
        ("n_lights", C.c_int32), ("lgt_pos", _pd), ("lgt_color", _pd), ("lgt_spec", _pd), ("lgt_shadow", _pd),
        ("lgt_radius", _pd),
        ("n_nodes", C.c_int32), ("node_lo", _pd), ("node_hi", _pd), ("node_right", _pi), ("node_first", _pi),
        ("node_count", _pi), ("n_leaf_refs", C.c_int32), ("leaf_prim", _pi),
        ("cam_pos", _d3), ("cam_x", _d3), ("cam_y", _d3), ("cam_z", _d3),
        ("cam_screen_distance", C.c_double), ("cam_screen_width", C.c_double),
        ("background", _d3), ("sh_rays", C.c_int32), ("rec_max", C.c_int32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("sh_rays", C.c_int32), ("rec_max", C.c_int32),
        ("jitter", C.c_int32), ("seed", C.c_uint32), ("rank", C.c_int32), ("world", C.c_int32),
        ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_int64), ("shadow_rays", C.c_int64), ("reflect_rays", C.c_int64),
        ("transmit_rays", C.c_int64), ("hits", C.c_int64), ("levels", C.c_int64), ("max_queue", C.c_int64),
        ("ms_total", C.c_double), ("ms_intersect", C.c_double), ("ms_shadow", C.c_double), ("ms_shade", C.c_double),
        ("ms_gen", C.c_double), ("ms_resolve", C.c_double), ("launches_per_family", C.c_int64),
        ("ms_beam_setup", C.c_double), ("beam_setup_launches", C.c_int64),
    ]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["rays"] = d["primary_rays"] + d["shadow_rays"] + d["reflect_rays"] + d["transmit_rays"]
        return d


class BspStats(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double), ("ms_device", C.c_double), ("n_nodes", C.c_int32), ("n_leaf_refs", C.c_int32),
        ("depth", C.c_int32), ("launches", C.c_int32), ("ref_visits", C.c_int64), ("bytes_scratch", C.c_int64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [
        ("lines", C.c_int64), ("n_records", C.c_int64), ("n_finite", C.c_int64), ("threads", C.c_int32),
        ("has_camera", C.c_int32), ("has_settings", C.c_int32), ("reserved", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


# rt_scene_desc tables: name -> (count field, columns)
SCENE_TABLES = {
    "surf_kind": ("n_surfaces", 1), "surf_geom": ("n_surfaces", 1), "surf_mat": ("n_surfaces", 1),
    "sph_center": ("n_spheres", 3), "sph_radius": ("n_spheres", 1), "sph_r2": ("n_spheres", 1), "sph_id": ("n_spheres", 1),
    "box_center": ("n_boxes", 3), "box_scale": ("n_boxes", 1), "box_lo": ("n_boxes", 3), "box_hi": ("n_boxes", 3),
    "box_id": ("n_boxes", 1),
    "pln_normal": ("n_planes", 3), "pln_d": ("n_planes", 1), "pln_id": ("n_planes", 1),
    "mat_kd": ("n_materials", 3), "mat_ks": ("n_materials", 3), "mat_kr": ("n_materials", 3),
    "mat_phong": ("n_materials", 1), "mat_transparency": ("n_materials", 1),
    "lgt_pos": ("n_lights", 3), "lgt_color": ("n_lights", 3), "lgt_spec": ("n_lights", 1), "lgt_shadow": ("n_lights", 1),
    "lgt_radius": ("n_lights", 1),
    "node_lo": ("n_nodes", 3), "node_hi": ("n_nodes", 3), "node_right": ("n_nodes", 1), "node_first": ("n_nodes", 1),
    "node_count": ("n_nodes", 1), "leaf_prim": ("n_leaf_refs", 1),
}


def render_params(width, height, sh_rays=None, rec_max=None, jitter=False, seed=0, rank=0, world=1, tile_w=0,
                  tile_h=0, flags=0):
    return RenderParams(int(width), int(height), RT_USE_SCENE_VALUE if sh_rays is None else int(sh_rays),
                        RT_USE_SCENE_VALUE if rec_max is None else int(rec_max), 1 if jitter else 0,
                        int(seed) & 0xFFFFFFFF, int(rank), int(world), int(tile_w), int(tile_h), int(flags), 0)
