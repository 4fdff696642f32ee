"""Host logic without a GPU: scene parsing, camera basis, the restated BSP build, scene packing,
tile layout arithmetic and the C-ABI library's exported surface."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, load_scene, scene_path
from ray_tracing_amd import abi, native


@pytest.mark.parametrize("tag", ["pool", "original", "small", "synth400"])
def test_bsp_build_matches_reference_tree(tag):
    g, meta = golden("bsp_" + tag)
    sc = load_scene(tag)
    f = sc["bsp"].flat
    assert len(f.right) == meta["nodes"]
    assert np.array_equal(f.lo, g["lo"]) and np.array_equal(f.hi, g["hi"])
    assert np.array_equal((f.right < 0).astype(np.int8), g["is_leaf"])
    assert np.array_equal(f.first, g["first"]) and np.array_equal(f.count, g["count"])
    ids = np.array([sc["bsp"].finite_objects[k].index for k in f.leaf_prim])
    assert np.array_equal(ids, g["leaf_ids"])
    assert [p.index for p in sc["bsp"].inf_plains] == list(g["root_planes"])
    # object view of the same tree (reference attribute names)
    node = sc["bsp"]
    while node.left is not None:
        assert node.surfaces is None and node.cut[0] in "xyz"
        node = node.left
    assert [s.surface.index for s in node.surfaces] == list(ids[:len(node.surfaces)])


def test_parser_index_rules_and_camera_basis():
    sc = load_scene("original")
    surfaces, lights, mats = sc["surfaces"], sc["lights"], sc["materials"]
    assert type(surfaces[0]).__name__ == "Background" and surfaces[0].index == 0
    assert all(s.index == k for k, s in enumerate(surfaces))
    assert [l.index for l in lights] == list(range(len(surfaces), len(surfaces) + len(lights)))   # shared counter
    assert [m.index for m in mats] == list(range(len(mats))) and mats[0].index == 0
    g, _ = golden("rays_original_12x12")
    cam = sc["camera"]
    assert np.array_equal(cam.x_dir, g["cam_x"]) and np.array_equal(cam.y_dir, g["cam_y"])
    assert np.array_equal(cam.z_dir, g["cam_z"])
    with pytest.raises(ValueError):
        from Parser import parse_scene_file
        bad = os.path.join(ROOT, "tests", "golden", "_bad_scene.txt")
        with open(bad, "w") as f:
            f.write("cam 0 0 0 0 0 1 0 1 0 1 1\nfoo 1 2 3\n")
        try:
            parse_scene_file(bad)
        finally:
            os.unlink(bad)


def test_scene_pack_tables():
    g, _ = golden("shade_terms")
    from Material import Material
    from ray_tracing_amd import scene_pack
    from surfaces.Background import Background
    from surfaces.Cube import Cube
    from surfaces.InfinitePlane import InfinitePlane
    from surfaces.Sphere import Sphere
    mats = [Material([.1, .2, .3], [.4, .5, .6], [0, 0, 0], float(g["mat_shininess_in"][k]),
                     float(g["mat_transparency_in"][k]), k) for k in range(3)]
    surfaces = [Background(), Sphere([0, 0, 0], 0.3, 1, 1), InfinitePlane([0, 2, 0], -1, 2, 2),
                Cube(np.array([1.0, 0.0, 0.0]), 0.13, 0, 3), Sphere([2, 0, 0], 0.7, 2, 4)]
    ps = scene_pack.pack_scene(surfaces, mats)
    a = ps.arrays
    # integer shininess: truncation toward zero, as Material.get_materials_base_colors does (Material.py:49,58)
    assert list(a["mat_phong"]) == [int(x) for x in g["mat_phong"][[np.argmax(g["mat_index"] == k) for k in range(3)]]]
    assert list(a["surf_kind"]) == [0, 1, 3, 2, 1] and list(a["surf_geom"]) == [-1, 0, 0, 0, 1]
    assert list(a["surf_mat"]) == [0, 1, 2, 0, 2]
    assert a["sph_r2"][0] == 0.3 ** 2 and a["sph_r2"][1] == 0.7 ** 2
    assert np.array_equal(a["box_lo"][0], np.array([1.0, 0, 0]) - 0.13 / 2)
    assert np.array_equal(a["pln_normal"][0], [0, 1, 0]) and a["pln_d"][0] == 0.5
    assert ps.desc.far_clip == 1 and ps.desc.n_nodes == 1 and list(a["leaf_prim"]) == [0, 2, 1]


def test_tile_layout_partitions_every_pixel_once():
    for w, h, world in ((1920, 1080, 8), (100, 52, 4), (17, 9, 3), (64, 64, 2)):
        seen = np.zeros(w * h, dtype=int)
        sizes = set()
        for r in range(world):
            pix = native.tile_layout(abi.render_params(w, h, rank=r, world=world))
            sizes.add(len(pix))
            np.add.at(seen, pix[pix >= 0], 1)
        assert len(sizes) == 1 and seen.min() == 1 and seen.max() == 1
    assert np.array_equal(native.tile_layout(abi.render_params(5, 3)), np.arange(15))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "rt_amd.h")) as f:
        header = f.read()
    declared = set(re.findall(r"^(?:int|int64_t|void|const char \*|const rt_scene_desc \*)\s*\*?\s*(rt_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert {"rt_create", "rt_render", "rt_intersect", "rt_trace", "rt_unit_prim_intersect", "rt_bsp_build",
            "rt_scene_parse", "rt_scene_get_desc", "rt_scene_free"} <= declared
    lib = native.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(native.EXPORTS)
    assert lib.rt_abi_version() == abi.RT_AMD_ABI_VERSION


def test_ctypes_structs_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rt_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(rt_scene_desc),sizeof(rt_render_params),sizeof(rt_stats),offsetof(rt_scene_desc,cam_pos),'
                   'offsetof(rt_scene_desc,leaf_prim),offsetof(rt_stats,ms_total));'
                   'printf("%zu %zu\\n",sizeof(rt_bsp_stats),offsetof(rt_bsp_stats,ref_visits));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(abi.SceneDesc), ctypes.sizeof(abi.RenderParams), ctypes.sizeof(abi.Stats),
            abi.SceneDesc.cam_pos.offset, abi.SceneDesc.leaf_prim.offset, abi.Stats.ms_total.offset,
            ctypes.sizeof(abi.BspStats), abi.BspStats.ref_visits.offset]
    assert got == want


def test_flag_constants_match_the_header():
    with open(os.path.join(ROOT, "include", "rt_amd.h")) as f:
        header = f.read()
    flags = dict(re.findall(r"^#define (RT_FLAG_[A-Z_]+) (\d+)", header, flags=re.M))
    assert set(flags) == {"RT_FLAG_EXACT_TRAVERSAL", "RT_FLAG_PROFILE", "RT_FLAG_NO_BEAM", "RT_FLAG_NO_OVERLAP", "RT_FLAG_NO_SMALL", "RT_FLAG_TIME_DOMINANT", "RT_FLAG_NO_TILES"}
    for name, value in flags.items():
        assert getattr(abi, name) == int(value), name
    values = sorted(int(v) for v in flags.values())
    assert values == [1, 2, 4, 8, 16, 32, 64]


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(native.NativeError) as e:
        native.Context(load_scene("pool")["packed"])
    assert e.value.code == -2


def test_tree_build_without_a_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(native.NativeError) as e:
        native.bsp_build(np.zeros((4, 3)), np.ones((4, 3)))
    assert e.value.code == -2
    from BSPNode import BSPNode
    with pytest.raises(native.NativeError):
        BSPNode.build_bsp_tree_device(load_scene("pool")["surfaces"])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ray-tracing_amd")
    offenders = []
    for base, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(base, fn)).read()
                if re.search(r"\borc\b|liboracle|oracle/", text) and "oracle/oracle.cpp" not in text:
                    offenders.append(fn)
                elif re.search(r"import orc|liboracle", text):
                    offenders.append(fn)
    text = open(os.path.join(ROOT, "ray_tracer.py")).read()
    assert "orc" not in re.findall(r"import (\w+)", text)
    assert not offenders, offenders
