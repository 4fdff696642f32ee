"""rt_scene_parse (csrc/rt_scene.cpp) against the Python path it replaces: Parser.parse_scene_file +
scene_pack.pack_scene, whose object constructors are pinned to the reference's by tests/test_host.py and
tests/test_oracle.py.  Every table and scalar must be bit-identical.  No GPU needed (the tree is handed in)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, load_scene, scene_path
from ray_tracing_amd import abi, native, scene_pack

sys.path.insert(0, os.path.join(ROOT, "tools"))

SCALARS = ("cam_screen_distance", "cam_screen_width", "sh_rays", "rec_max", "far_clip", "n_surfaces", "n_spheres",
           "n_boxes", "n_planes", "n_materials", "n_lights", "n_nodes", "n_leaf_refs")
VECTORS = ("cam_pos", "cam_x", "cam_y", "cam_z", "background")


def assert_same_tables(ns, packed, tree=True):
    a, b = ns.arrays, packed.arrays
    for name in abi.SCENE_TABLES:
        if not tree and (name.startswith("node_") or name == "leaf_prim"):
            continue
        x, y = np.asarray(a[name]).reshape(-1), np.asarray(b[name]).reshape(-1)
        assert x.dtype == y.dtype and np.array_equal(x, y, equal_nan=x.dtype == np.float64), name
    for f in SCALARS:
        if tree or f not in ("n_nodes", "n_leaf_refs"):
            assert getattr(ns.desc, f) == getattr(packed.desc, f), f
    for f in VECTORS:
        assert list(getattr(ns.desc, f)) == list(getattr(packed.desc, f)), f


def python_path(path):
    from BSPNode import BSPNode
    from Parser import parse_scene_file
    cam, st, _o, surfaces, mats, lights = parse_scene_file(path)
    bsp = BSPNode.build_bsp_tree(surfaces)
    return dict(camera=cam, settings=st, surfaces=surfaces, materials=mats, lights=lights, bsp=bsp,
                packed=scene_pack.pack_scene(surfaces, mats, lights, cam, st, bsp))


@pytest.mark.parametrize("tag", ["pool", "original", "small", "synth400"])
def test_fixture_scenes_give_identical_tables(tag):
    sc = load_scene(tag)
    ns = native.NativeScene(path=scene_path(tag))
    assert ns.desc.n_nodes == 0
    assert_same_tables(ns, sc["packed"], tree=False)
    ns.set_tree(sc["bsp"].flat)
    assert_same_tables(ns, sc["packed"])
    # index rules (Parser.py:31,54-66): lights share the surfaces' counter, materials count from 1 after the background
    lights, mats = ns.ids()
    assert list(lights) == [l.index for l in sc["lights"]] and list(mats) == [m.index for m in sc["materials"]]
    small, big = ns.corners()
    fin = sc["bsp"].finite_objects
    assert np.array_equal(small, np.array([o.get_enclosing_values()[0] for o in fin]))
    assert np.array_equal(big, np.array([o.get_enclosing_values()[1] for o in fin]))
    # from memory, as str and as bytes
    text = open(scene_path(tag)).read()
    for t in (text, text.encode()):
        ns2 = native.NativeScene(text=t)
        assert_same_tables(ns2, sc["packed"], tree=False)


def _random_scene_text(seed):
    rng = np.random.default_rng(seed)

    def num(x):
        style = rng.integers(0, 6)
        if style == 0:
            return repr(float(x))
        if style == 1:
            return "%.17g" % x
        if style == 2:
            return "%+.9e" % x
        if style == 3:
            return "%.3f" % x
        if style == 4 and float(x).is_integer():
            return "%d" % x
        return "%r" % float(np.float32(x))

    sep = lambda: [" ", "  ", "\t", " \t "][rng.integers(0, 4)]  # noqa: E731
    lines = ["# random scene %d" % seed, ""]
    pos, look = rng.normal(size=3) * 10, rng.normal(size=3) * 3
    up = rng.normal(size=3)
    lines.append("cam" + sep() + sep().join(num(v) for v in [*pos, *look, *up, rng.uniform(0.5, 3), rng.uniform(0.5, 3)]))
    if seed % 3:
        lines.append("set" + sep() + sep().join(num(v) for v in [*rng.uniform(0, 1, 3), rng.integers(1, 6), rng.integers(0, 8)]))
    n_mat = int(rng.integers(1, 9))
    body = []
    for _ in range(n_mat):
        body.append("mtl" + sep() + sep().join(num(v) for v in [*rng.uniform(0, 1, 9), rng.uniform(0, 90), rng.uniform(0, 1)]))
    for _ in range(int(rng.integers(20, 400))):
        k = rng.integers(0, 10)
        m = int(rng.integers(0, n_mat + 1))
        if k < 4:
            body.append("sph" + sep() + sep().join(num(v) for v in [*rng.normal(size=3) * 8, rng.uniform(0.05, 1.5)]) + sep() + str(m))
        elif k < 7:
            body.append("box" + sep() + sep().join(num(v) for v in [*rng.normal(size=3) * 8, rng.uniform(0.05, 2.5)]) + sep() + str(m))
        elif k < 9:
            body.append("pln" + sep() + sep().join(num(v) for v in [*rng.normal(size=3) * 10 ** rng.uniform(-2, 2),
                                                                  rng.normal() * 5]) + sep() + str(m))
        else:
            body.append("lgt" + sep() + sep().join(num(v) for v in [*rng.normal(size=3) * 20, *rng.uniform(0, 1, 3),
                                                                  rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 3)]))
        if rng.integers(0, 15) == 0:
            body.append(["", "   ", "# note", "\t#x 1 2"][rng.integers(0, 4)])
    rng.shuffle(body)
    eol = ["\n", "\r\n"][seed % 2]
    return eol.join(lines + ["  " + b if rng.integers(0, 9) == 0 else b for b in body]) + (eol if seed % 5 else "")


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_files_give_identical_tables(seed, tmp_path):
    text = _random_scene_text(seed)
    p = tmp_path / "scene.txt"
    p.write_bytes(text.encode())
    py = python_path(str(p))
    ns = native.NativeScene(path=str(p))
    ns.set_tree(py["bsp"].flat)
    assert_same_tables(ns, py["packed"])
    lights, mats = ns.ids()
    assert list(lights) == [l.index for l in py["lights"]] and list(mats) == [m.index for m in py["materials"]]
    assert ns.info["has_settings"] == (1 if seed % 3 else 0)


def test_camera_and_plane_constants_on_many_random_vectors():
    """The two places where NumPy's 1-D norm (a BLAS dot product) enters: Camera.look_direction and the plane normal."""
    from Camera import Camera
    from surfaces.InfinitePlane import InfinitePlane
    rng = np.random.default_rng(3)
    lines, cams, planes = [], [], []
    for k in range(3000):
        n, off = rng.normal(size=3) * 10 ** rng.uniform(-3, 3), rng.normal() * 7
        planes.append(InfinitePlane([float(x) for x in n], float(off), 1, k + 1))
        lines.append("pln " + " ".join(repr(float(x)) for x in [*n, off]) + " 1")
    ns = native.NativeScene(text="\n".join(lines))
    a = ns.arrays
    assert np.array_equal(a["pln_normal"], np.array([p.normal for p in planes]))
    assert np.array_equal(a["pln_d"], np.array([p.offset for p in planes]))
    for k in range(500):
        v = [float(x) for x in rng.normal(size=9) * 10 ** rng.uniform(-1, 2)] + [1.4, 1.0]
        cam = Camera(v[:3], v[3:6], v[6:9], v[9], v[10])
        d = native.NativeScene(text="cam " + " ".join(repr(x) for x in v)).desc
        assert list(d.cam_x) == list(cam.x_dir) and list(d.cam_y) == list(cam.y_dir) and list(d.cam_z) == list(cam.z_dir)


def test_threads_cut_the_text_at_line_ends():
    import gen_scene
    text = gen_scene.synth_mixed(200, 77, 3, 4).replace("\n", "\r\n", 5000)      # 40k primitives, > 3 MB
    one = native.NativeScene(text=text, threads=1)
    many = native.NativeScene(text=text, threads=7)
    assert one.info["threads"] == 1 and many.info["threads"] > 1
    assert many.info["lines"] == one.info["lines"] == text.count("\n") + (0 if text.endswith("\n") else 1)
    a, b = one.arrays, many.arrays
    for name in abi.SCENE_TABLES:
        assert np.array_equal(a[name], b[name]), name
    sa, sb = one.corners(), many.corners()
    assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])
    # and against the Python objects on the same text
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False, newline="") as f:
        f.write(text)
    try:
        py = python_path(f.name)
    finally:
        os.unlink(f.name)
    many.set_tree(py["bsp"].flat)
    assert_same_tables(many, py["packed"])


@pytest.mark.parametrize("text,needle", [
    ("cam 0 0 0 0 0 1 0 1 0 1 1\nfoo 1 2 3\n", "line 2: Unknown object type: foo"),
    ("sph 0 0 0 1 1\n\n# c\nsph 0 0 zero 1 1\n", "line 4: could not convert string to float: 'zero'"),
    ("sph 0 0 0 1\n", "line 1"),
    ("sph 0 0 0 0x10 1\n", "line 1"),
    ("sphere 0 0 0 1 1\n", "Unknown object type: sphere"),
])
def test_bad_lines_raise_value_error_with_the_line_number(text, needle):
    with pytest.raises(ValueError) as e:
        native.NativeScene(text=text)
    assert needle in str(e.value)


def test_number_forms_python_accepts():
    ns = native.NativeScene(text="sph +1.5 -.5 5. 1e0 2\nsph 1E+1 -0 0.0 inf 1\nbox nan 0 0 1e-400 1 99 98\n")
    a = ns.arrays
    assert np.array_equal(a["sph_center"], [[1.5, -0.5, 5.0], [10.0, -0.0, 0.0]]) and np.signbit(a["sph_center"][1, 1])
    assert a["sph_radius"][1] == np.inf and np.isnan(a["box_center"][0, 0]) and a["box_scale"][0] == 0.0
    assert list(a["surf_mat"]) == [0, 2, 1, 1] and list(a["sph_id"]) == [1, 2] and list(a["box_id"]) == [3]


def test_oracle_frame_from_native_tables_matches_the_reference_frame(oracle):
    g, meta = golden("frame_pool_64x36_s1_nojitter")
    sc = load_scene("pool")
    ns = native.NativeScene(path=scene_path("pool"))
    ns.set_tree(sc["bsp"].flat)
    w, h = meta["width"], meta["height"]
    r = oracle.render(ns, abi.render_params(w, h, sh_rays=meta["sh_rays"], rec_max=meta["rec_max"], jitter=False))
    assert np.array_equal(r["rgb8"].reshape(h, w, 3), g["rgb8_q4fix"])
