"""The bench workload at its full size (BASELINE.json configs[2]: synthetic 10 k-sphere scene, 1920x1080, 5x5 shadow
rays, recursion 5, jitter on).  The reference cannot render it (memory), so parity at this size goes through
properties that do not depend on the size:

  * the oracle renders single ranks of a 48-way tile partition of the SAME frame (1/48 of the pixels, same jitter
    hash): ids, ray counters and the 8-bit image of those tiles must agree with the GPU's;
  * the fast path (float32 culling, beam verdicts, filtered predicates, pipelined levels) must give the frame of the
    float64 reference-order walk, bit for bit;
  * the 8 ranks' tiles of a partitioned frame reassemble to the single-GPU frame, bit for bit.
"""
import os
import sys

import numpy as np
import pytest

from ray_tracing_amd import abi, native

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SH, REC, SEED = 1920, 1080, 5, 5, 1003


@pytest.fixture(scope="module")
def bench_scene():
    sys.path.insert(0, ROOT)
    import bench
    scene, _ = bench.build_scene("synth10k")
    ctx = native.Context(scene.packed, device=0)
    yield scene, ctx
    ctx.close()


def params(**kw):
    return abi.render_params(W, H, sh_rays=SH, rec_max=REC, jitter=True, seed=SEED, **kw)


@pytest.fixture(scope="module")
def full_frame(bench_scene):
    _, ctx = bench_scene
    return ctx.render(params())


@pytest.mark.parametrize("rank", [0, 29])
def test_tiles_of_the_full_frame_match_the_oracle(bench_scene, full_frame, oracle, rank):
    scene, ctx = bench_scene
    p = params(rank=rank, world=48)
    out = ctx.render(p)
    ref = oracle.render(scene.packed, p, threads=min(16, os.cpu_count() or 1))
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
        assert out["stats"][key] == ref["counters"][key], key
    assert out["stats"]["shadow_rays"] > 5_000_000
    # and these tiles are the ones of the unpartitioned frame
    idx = native.tile_layout(p)
    ok = idx >= 0
    assert np.array_equal(out["rgb8"].reshape(-1, 3)[ok], full_frame["rgb8"].reshape(-1, 3)[idx[ok]])


def test_full_frame_equals_the_exact_walk(bench_scene, full_frame):
    _, ctx = bench_scene
    exact = ctx.render(params(flags=abi.RT_FLAG_EXACT_TRAVERSAL))
    assert np.array_equal(full_frame["ids"], exact["ids"])
    assert full_frame["stats"]["rays"] == exact["stats"]["rays"] > 270_000_000
    assert np.array_equal(full_frame["rgb"], exact["rgb"])
    assert np.array_equal(full_frame["rgb8"], exact["rgb8"])


def test_eight_ranks_reassemble_the_full_frame(bench_scene, full_frame):
    _, ctx = bench_scene
    image = np.zeros((W * H, 3), dtype=np.uint8)
    rays = 0
    for r in range(8):
        p = params(rank=r, world=8)
        out = ctx.render(p, want_ids=False)
        idx = native.tile_layout(p)
        ok = idx >= 0
        image[idx[ok]] = out["rgb8"].reshape(-1, 3)[ok]
        rays += out["stats"]["rays"]
    assert np.array_equal(image, full_frame["rgb8"].reshape(-1, 3))
    assert rays == full_frame["stats"]["rays"]


# ---- BASELINE.json configs[4]: one million mixed spheres / boxes with transparency and reflection, recursion 10 ----
@pytest.fixture(scope="module")
def lattice_scene():
    sys.path.insert(0, ROOT)
    import bench
    scene, _ = bench.build_scene("synth1M")
    ctx = native.Context(scene.packed, device=0)
    yield scene, ctx
    ctx.close()


def test_million_primitive_scene_fast_path_equals_exact_walk_and_oracle(lattice_scene, oracle):
    """Reduced resolution, full scene: the long-beam templates, k_shadow_walk and the float64 queue all carry rays
    here.  Default path == float64 reference-order walk bit for bit; one rank of an 8-way partition == oracle."""
    scene, ctx = lattice_scene
    assert scene.packed.desc.n_spheres + scene.packed.desc.n_boxes == 1_000_000
    w, h = 320, 180
    kw = dict(sh_rays=3, rec_max=10, jitter=True, seed=77)
    fast = ctx.render(abi.render_params(w, h, **kw))
    exact = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **kw))
    assert np.array_equal(fast["ids"], exact["ids"])
    assert fast["stats"]["rays"] == exact["stats"]["rays"]
    assert fast["stats"]["transmit_rays"] > 0 and fast["stats"]["reflect_rays"] > 0
    assert np.array_equal(fast["rgb"], exact["rgb"])
    p = abi.render_params(w, h, rank=3, world=8, **kw)
    out = ctx.render(p)
    ref = oracle.render(scene.packed, p, threads=min(16, os.cpu_count() or 1))
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
        assert out["stats"][key] == ref["counters"][key], key


# ---- every error bound of the float32 filters has to scale with the scene ----
@pytest.mark.parametrize("tag,k", [("small", 0.001), ("small", 1000.0), ("original", 0.03), ("synth400", 40.0),
                                   ("pool", 1000.0), ("synth400", 0.001)])
def test_scaled_scenes_match_exact_walk_and_oracle(oracle, tag, k):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from check_beam_vs_exact import scaled
    from conftest import scene_path
    from ray_tracing_amd.render import FastScene
    sc = FastScene(text=scaled(open(scene_path(tag)).read(), k))
    with native.Context(sc.packed, device=0) as ctx:
        p = dict(sh_rays=3, rec_max=6, jitter=True, seed=3)
        fast = ctx.render(abi.render_params(240, 136, **p))
        exact = ctx.render(abi.render_params(240, 136, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **p))
    ref = oracle.render(sc.packed, abi.render_params(240, 136, **p), threads=min(16, os.cpu_count() or 1))
    assert np.array_equal(fast["ids"], exact["ids"]) and np.array_equal(fast["rgb"], exact["rgb"])
    assert np.array_equal(fast["ids"], ref["ids"])
    assert np.abs(fast["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
        assert fast["stats"][key] == ref["counters"][key], key
