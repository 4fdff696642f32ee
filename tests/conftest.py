import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ray_tracing_amd  # noqa: E402,F401  (puts the host mirror on sys.path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(g["meta_json"]).decode()) if "meta_json" in g.files else {}
    return g, meta


def scene_path(tag):
    return os.path.join(GOLDEN, "scenes", tag + ".txt")


_SCENES = {}


def load_scene(tag):
    """-> dict(camera, settings, surfaces, materials, lights, bsp, packed) for a fixture scene (cached)."""
    if tag not in _SCENES:
        from BSPNode import BSPNode
        from Parser import parse_scene_file
        from ray_tracing_amd import scene_pack
        cam, st, _objs, surfaces, mats, lights = parse_scene_file(scene_path(tag))
        bsp = BSPNode.build_bsp_tree(surfaces)
        packed = scene_pack.pack_scene(surfaces, mats, lights, cam, st, bsp)
        _SCENES[tag] = dict(camera=cam, settings=st, surfaces=surfaces, materials=mats, lights=lights, bsp=bsp,
                            packed=packed)
    return _SCENES[tag]


@pytest.fixture(scope="session")
def oracle():
    import orc
    orc.lib()
    return orc


_CTX = {}


@pytest.fixture(scope="session")
def gpu_ctx():
    """factory: tag -> native.Context on cuda:0 (cached for the session)."""
    from ray_tracing_amd import native

    def get(tag):
        if tag not in _CTX:
            _CTX[tag] = native.Context(load_scene(tag)["packed"], device=0)
        return _CTX[tag]

    yield get
    for c in _CTX.values():
        c.close()
    _CTX.clear()
