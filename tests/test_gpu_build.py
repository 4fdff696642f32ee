"""rt_bsp_build (csrc/rt_build.hip) against the reference's own trees (tests/golden/bsp_*.npz, written by
tools/make_golden.py from BSPNode.build_bsp_tree of the reference) and against the recursive host
restatement on larger scenes.  Bit-exact: node bounds, child links, leaf membership AND leaf order."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, load_scene, scene_path
from ray_tracing_amd import abi, native

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _same_flat(a, b):
    assert len(a.right) == len(b.right)
    for name in ("lo", "hi", "right", "first", "count", "leaf_prim"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


@pytest.mark.parametrize("tag", ["pool", "original", "small", "synth400"])
def test_device_tree_equals_reference_tree(tag):
    from BSPNode import BSPNode
    g, meta = golden("bsp_" + tag)
    sc = load_scene(tag)
    root = BSPNode.build_bsp_tree_device(sc["surfaces"])
    f = root.flat
    assert len(f.right) == meta["nodes"] == root.build_stats["n_nodes"]
    assert np.array_equal(f.lo, g["lo"]) and np.array_equal(f.hi, g["hi"])
    assert np.array_equal((f.right < 0).astype(np.int8), g["is_leaf"])
    assert np.array_equal(f.first, g["first"]) and np.array_equal(f.count, g["count"])
    ids = np.array([root.finite_objects[k].index for k in f.leaf_prim])
    assert np.array_equal(ids, g["leaf_ids"])
    assert [p.index for p in root.inf_plains] == list(g["root_planes"])
    _same_flat(f, sc["bsp"].flat)
    # object view with the reference's attribute names, materialised from the flat arrays
    node, host = root, sc["bsp"]
    while node.left is not None:
        assert node.surfaces is None and node.cut[0] == host.cut[0] and node.cut[1] == host.cut[1]
        assert np.array_equal(node.right.small_bound, host.right.small_bound)
        node, host = node.left, host.left
    assert [s.surface.index for s in node.surfaces] == [s.surface.index for s in host.surfaces]


def _scene_objects(text, tmp_path):
    from Parser import parse_scene_file
    p = tmp_path / "scene.txt"
    p.write_text(text)
    return parse_scene_file(str(p))


@pytest.mark.parametrize("which", ["synth10k", "mixed3600", "mixed90k"])
def test_device_tree_equals_host_tree_on_synthetic_scenes(which, tmp_path):
    import gen_scene
    from BSPNode import BSPNode
    text = {"synth10k": lambda: gen_scene.synth_spheres(100, 1234, 5, 5),
            "mixed3600": lambda: gen_scene.synth_mixed(60, 99, 3, 4),
            "mixed90k": lambda: gen_scene.synth_mixed(300, 4321, 5, 10)}[which]()
    surfaces = _scene_objects(text, tmp_path)[3]
    host = BSPNode.build_bsp_tree(surfaces)
    dev = BSPNode.build_bsp_tree_device(surfaces)
    _same_flat(dev.flat, host.flat)
    st = dev.build_stats
    assert st["n_leaf_refs"] == len(host.flat.leaf_prim) and st["depth"] >= 2 and st["ref_visits"] >= st["n_leaf_refs"]


def test_given_root_bounds_are_used():
    from BSPNode import BSPNode
    sc = load_scene("synth400")
    lo, hi = np.array([-40.0, -3.0, -41.5]), np.array([37.0, 9.25, 44.0])
    host = BSPNode.build_bsp_tree(sc["surfaces"], lo, hi)
    dev = BSPNode.build_bsp_tree_device(sc["surfaces"], lo, hi)
    assert np.array_equal(dev.flat.lo[0], lo) and np.array_equal(dev.flat.hi[0], hi)
    _same_flat(dev.flat, host.flat)


@pytest.mark.parametrize("n", [1, 10, 11, 12, 257])
def test_small_counts_and_leaf_threshold(n):
    from BSPNode import BSPNode
    from surfaces.Cube import Cube
    from surfaces.Sphere import Sphere
    rng = np.random.default_rng(n)
    surfaces = []
    for k in range(n):
        c = rng.uniform(-3, 3, 3)
        surfaces.append(Sphere(c, float(rng.uniform(0.1, 0.8)), 1, k + 1) if k % 3 else
                        Cube(c, float(rng.uniform(0.2, 1.0)), 1, k + 1))
    host = BSPNode.build_bsp_tree(surfaces)
    dev = BSPNode.build_bsp_tree_device(surfaces)
    _same_flat(dev.flat, host.flat)
    if n <= 10:
        assert len(dev.flat.right) == 1 and dev.flat.count[0] == n


def test_cells_shrink_to_epsilon_where_many_objects_meet():
    """Twelve cubes, one or two per octant, that all reach 1e-7 past the origin: the cells around the origin keep
    all twelve and are cut until |big - small| < EPSILON (BSPNode.py:67), 66 levels deep."""
    from BSPNode import BSPNode
    from surfaces.Cube import Cube
    signs = [(a, b, c) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)]
    surfaces = []
    for k in range(12):
        scale = 1.0 if k < 8 else 0.7
        surfaces.append(Cube(np.array(signs[k % 8], dtype=np.float64) * (scale / 2 - 1e-7), scale, 1, k + 1))
    host = BSPNode.build_bsp_tree(surfaces)
    dev = BSPNode.build_bsp_tree_device(surfaces)
    _same_flat(dev.flat, host.flat)
    assert dev.build_stats["depth"] == 67 and len(dev.flat.right) == 1023


def test_random_corner_arrays_with_duplicates_and_ties():
    """Raw corner arrays on a coarse lattice: many corners fall exactly on cut locations (the <= / >= ties of
    BSPNode.py:80-82) and many boxes are identical."""
    from BSPNode import BSPNode, FlatTree  # noqa: F401
    from surfaces.Cube import Cube
    rng = np.random.default_rng(11)
    n = 3000
    centre = rng.integers(-40, 41, size=(n, 3)).astype(np.float64) * 0.5
    centre[-300:] = centre[:300]
    scale = rng.integers(1, 5, size=n).astype(np.float64) * 0.5
    scale[-300:] = scale[:300]
    surfaces = [Cube(centre[k], float(scale[k]), 1, k + 1) for k in range(n)]
    host = BSPNode.build_bsp_tree(surfaces)
    t = native.bsp_build(centre - scale[:, None] / 2, centre + scale[:, None] / 2)
    for name in ("lo", "hi", "right", "first", "count", "leaf_prim"):
        assert np.array_equal(t[name], getattr(host.flat, name)), name


def test_empty_input_is_an_error():
    with pytest.raises(native.NativeError) as e:
        native.bsp_build(np.zeros((0, 3)), np.zeros((0, 3)))
    assert e.value.code == -1


def test_frame_from_a_device_built_scene_matches_the_reference_frame():
    from ray_tracing_amd.render import Renderer, Scene
    g, meta = golden("frame_pool_160x90_s5_nojitter")
    r = Renderer(Scene.from_file(scene_path("pool"), build="device"))
    try:
        out = r.frame(meta["width"], meta["height"], sh_rays=meta["sh_rays"], rec_max=meta["rec_max"], jitter=False)
    finally:
        r.close()
    assert np.abs(out["rgb8"].astype(int) - g["rgb8_q4fix"].astype(int)).max() <= 1
    assert abi.RT_OK == 0


# ------------------------------------------------------------------ text -> tables -> tree -> frame, no Python objects
@pytest.mark.parametrize("tag", ["pool", "original", "small", "synth400"])
def test_fast_scene_has_the_tables_and_tree_of_the_object_path(tag):
    from ray_tracing_amd.render import FastScene
    from test_scene_native import assert_same_tables
    sc = load_scene(tag)
    fs = FastScene.from_file(scene_path(tag))
    assert_same_tables(fs.packed, sc["packed"])
    assert fs.build_stats["n_nodes"] == sc["packed"].desc.n_nodes
    assert fs.settings.root_number_shadow_rays == sc["settings"].root_number_shadow_rays
    assert fs.settings.max_recursions == sc["settings"].max_recursions


def test_fast_scene_on_a_mixed_lattice(tmp_path):
    import gen_scene
    from ray_tracing_amd.render import FastScene
    from test_scene_native import assert_same_tables, python_path
    p = tmp_path / "mixed.txt"
    p.write_text(gen_scene.synth_mixed(60, 99, 3, 4))
    py = python_path(str(p))
    fs = FastScene.from_file(str(p))
    assert_same_tables(fs.packed, py["packed"])
    a = native.Context(fs.packed)
    b = native.Context(py["packed"])
    try:
        prm = abi.render_params(160, 90, jitter=True, seed=5)
        ra, rb = a.render(prm), b.render(prm)
    finally:
        a.close()
        b.close()
    assert np.array_equal(ra["ids"], rb["ids"]) and np.array_equal(ra["rgb8"], rb["rgb8"])


def test_main_writes_the_same_image_with_either_ingestion(tmp_path):
    import ray_tracer
    from PIL import Image
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    common = [scene_path("pool"), None, "--width", "96", "--height", "54", "--no-jitter", "--shadow-rays", "2"]
    ray_tracer.main([common[0], a] + common[2:])
    ray_tracer.main([common[0], b] + common[2:] + ["--python-ingest"])
    ia, ib = np.asarray(Image.open(a)), np.asarray(Image.open(b))
    assert ia.shape == (54, 96, 3) and np.array_equal(ia, ib) and ia.max() > 0


def test_a_tree_that_blows_up_is_refused_not_built():
    """more than ten objects through one point make the reference's builder copy them into both children level
    after level until the cells fall below EPSILON - 2^60 references for twelve unit cubes on one spot (the reference dies
    of it too).  rt_bsp_build stops when a level would exceed 2^30 references and says so: RT_ERR_UNSUPPORTED with a
    message, no crash, no endless build; the next call works."""
    n = 12
    lo, hi = np.full((n, 3), -0.5), np.full((n, 3), 0.5)
    with pytest.raises(native.NativeError) as e:
        native.bsp_build(lo, hi)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED
    assert "2^30" in str(e.value)
    t = native.bsp_build(lo[:10], hi[:10])          # ten of them: one leaf (BSPNode.py:67)
    assert len(t["right"]) == 1 and t["count"][0] == 10
