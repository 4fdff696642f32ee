"""The bench workload at its full size (BASELINE.json configs[2]: synthetic 10 k-sphere scene, 1920x1080, 5x5 shadow
rays, recursion 5, jitter on).  The reference cannot render it (memory), so parity at this size goes through
properties that do not depend on the size:

  * the oracle renders the whole frame once (all 2 M pixels, ~10 s of the box's host cores) and single ranks of a 48-way tile
    partition of the SAME frame (1/48 of the pixels, same jitter hash): ids, ray counters and the 8-bit image must agree
    with the GPU's;
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


def test_the_whole_full_size_frame_matches_the_oracle(bench_scene, full_frame, oracle):
    """All 2 073 600 pixels of the bench frame (275.6 M rays) against the CPU oracle on the host's cores (~10 s on the GPU box's
    16): ids bit-exact, 8-bit image within 1, every ray counter equal."""
    scene, _ = bench_scene
    ref = oracle.render(scene.packed, params(), threads=min(16, os.cpu_count() or 1), want_f64=False)
    assert np.array_equal(full_frame["ids"], ref["ids"])
    diff = np.abs(full_frame["rgb8"].astype(np.int16) - ref["rgb8"].astype(np.int16))
    assert diff.max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits", "rays"):
        assert full_frame["stats"][key] == ref["counters"][key], key
    assert ref["counters"]["rays"] > 270_000_000


def test_the_whole_3840x2160_frame_matches_the_oracle(bench_scene, oracle):
    """BASELINE.json configs[3], what `bench.py --gpus N` splits over its ranks: 8.3 M pixels, 1.1 G rays, ~35 s of the box's host
    cores for the oracle."""
    scene, ctx = bench_scene
    p = abi.render_params(3840, 2160, sh_rays=SH, rec_max=REC, jitter=True, seed=SEED)
    out = ctx.render(p, want_rgb=False)
    ref = oracle.render(scene.packed, p, threads=min(16, os.cpu_count() or 1), want_f64=False)
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(np.int16) - ref["rgb8"].astype(np.int16)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits", "rays"):
        assert out["stats"][key] == ref["counters"][key], key
    assert ref["counters"]["rays"] > 1_000_000_000


@pytest.mark.parametrize("workload,w,h", [("pool", 1920, 1080), ("original", 500, 500)])
def test_whole_frames_of_the_reference_scenes_at_their_own_settings_match_the_oracle(oracle, workload, w, h):
    """pool.txt (BASELINE.json configs[1]) and original.txt (configs[0]) with the shadow rays and recursion depth their files ask
    for and hashed jitter: the frames `extra_frames` of the bench line times.  (The reference's own frames of these scenes, jitter
    off, are in tests/test_gpu_parity.py.)"""
    sys.path.insert(0, ROOT)
    import bench
    scene, _ = bench.build_scene(workload)
    p = abi.render_params(w, h, jitter=True, seed=77)
    with native.Context(scene.packed, device=0) as ctx:
        out = ctx.render(p, want_rgb=False)
    ref = oracle.render(scene.packed, p, threads=min(16, os.cpu_count() or 1), want_f64=False)
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(np.int16) - ref["rgb8"].astype(np.int16)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits", "rays"):
        assert out["stats"][key] == ref["counters"][key], key


def test_full_frame_equals_the_exact_walk(bench_scene, full_frame):
    _, ctx = bench_scene
    exact = ctx.render(params(flags=abi.RT_FLAG_EXACT_TRAVERSAL))
    assert np.array_equal(full_frame["ids"], exact["ids"])
    assert full_frame["stats"]["rays"] == exact["stats"]["rays"] > 270_000_000
    assert np.array_equal(full_frame["rgb"], exact["rgb"])
    assert np.array_equal(full_frame["rgb8"], exact["rgb8"])


def test_full_frame_takes_the_tile_beams_and_equals_the_pairs_own_walks(bench_scene, full_frame):
    """At this size level 0 takes its candidates from the binned rows of its 8 x 8 pixel tiles; RT_FLAG_NO_TILES sends every
    pair through the tree itself.  Same frame and counters; also at 3840x2160 (BASELINE.json configs[3], whole frame)."""
    _, ctx = bench_scene
    lib = native.lib()
    n0 = lib.rt_debug_tile_frames(ctx._h)
    again = ctx.render(params())
    assert lib.rt_debug_tile_frames(ctx._h) == n0 + 1, "the full-size frame did not take the tile beams"
    plain = ctx.render(params(flags=abi.RT_FLAG_NO_TILES))
    assert lib.rt_debug_tile_frames(ctx._h) == n0 + 1
    for out in (again, plain):
        assert np.array_equal(out["ids"], full_frame["ids"]) and np.array_equal(out["rgb"], full_frame["rgb"])
        assert out["stats"]["rays"] == full_frame["stats"]["rays"]
    kw = dict(sh_rays=SH, rec_max=REC, jitter=True, seed=SEED + 1)
    big = ctx.render(abi.render_params(3840, 2160, **kw), want_ids=False)
    big_plain = ctx.render(abi.render_params(3840, 2160, flags=abi.RT_FLAG_NO_TILES, **kw), want_ids=False)
    assert lib.rt_debug_tile_frames(ctx._h) == n0 + 2
    assert np.array_equal(big["rgb"], big_plain["rgb"]) and big["stats"]["rays"] == big_plain["stats"]["rays"] > 1_000_000_000
    # one of two ranks' share of that frame (8x8 tiles interleaved: a block's neighbours in slot order are two tiles apart)
    half = ctx.render(abi.render_params(3840, 2160, rank=1, world=2, **kw), want_ids=False)
    half_plain = ctx.render(abi.render_params(3840, 2160, rank=1, world=2, flags=abi.RT_FLAG_NO_TILES, **kw), want_ids=False)
    assert lib.rt_debug_tile_frames(ctx._h) == n0 + 3
    assert np.array_equal(half["rgb"], half_plain["rgb"]) and half["stats"]["rays"] == half_plain["stats"]["rays"]


def test_tile_beams_on_a_mixed_lattice_equal_the_pairs_own_walks():
    """The box-capable template of the tile-beam set-up at a size where the default path takes it: 10 k spheres and boxes
    with transparency and reflection (tools/gen_scene.synth_mixed(100)), 1920x1080, 5x5 shadow rays, recursion 10."""
    sys.path.insert(0, ROOT)
    import bench
    scene, _ = bench.build_scene("synth10kmix")
    with native.Context(scene.packed, device=0) as ctx:
        kw = dict(sh_rays=5, rec_max=10, jitter=True, seed=77)
        n0 = native.lib().rt_debug_tile_frames(ctx._h)
        tiles = ctx.render(abi.render_params(W, H, **kw))
        assert native.lib().rt_debug_tile_frames(ctx._h) == n0 + 1
        plain = ctx.render(abi.render_params(W, H, flags=abi.RT_FLAG_NO_TILES, **kw))
        assert np.array_equal(tiles["ids"], plain["ids"]) and np.array_equal(tiles["rgb"], plain["rgb"])
        assert tiles["stats"]["rays"] == plain["stats"]["rays"] > 250_000_000
        exact = ctx.render(abi.render_params(W, H, flags=abi.RT_FLAG_NO_BEAM, **kw))      # every shadow ray on its own, float64
        assert np.array_equal(tiles["rgb"], exact["rgb"]) and tiles["stats"]["rays"] == exact["stats"]["rays"]


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


# ---- the filter margins at full size and at the edges of their range ("pin the filter margins") ----
def test_million_primitive_scene_full_frame_equals_float64_shadows(lattice_scene):
    """BASELINE.json configs[4] at 1920x1080 with 5x5 shadow rays: the default path (beam verdicts, float32 filtered
    predicates, k_shadow_walk, long-beam templates) against RT_FLAG_NO_BEAM (every shadow ray decided by its own
    float64-confirmed walk).  The float32 images must be identical."""
    _, ctx = lattice_scene
    kw = dict(sh_rays=5, rec_max=10, jitter=True, seed=11)
    fast = ctx.render(abi.render_params(1920, 1080, **kw))
    exact = ctx.render(abi.render_params(1920, 1080, flags=abi.RT_FLAG_NO_BEAM, **kw))
    assert fast["stats"]["shadow_rays"] > 300_000_000
    assert np.array_equal(fast["ids"], exact["ids"])
    assert fast["stats"]["rays"] == exact["stats"]["rays"]
    assert int((fast["rgb"] != exact["rgb"]).any(-1).sum()) == 0


def test_million_primitive_scene_full_size_rank_matches_the_oracle(lattice_scene, oracle):
    """... and one rank's share of that 1920x1080 frame (an eighth of its tiles, 38 M rays) against the CPU oracle."""
    scene, ctx = lattice_scene
    p = abi.render_params(1920, 1080, sh_rays=5, rec_max=10, jitter=True, seed=11, rank=5, world=8)
    out = ctx.render(p, want_rgb=False)
    ref = oracle.render(scene.packed, p, threads=min(16, os.cpu_count() or 1), want_f64=False)
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(np.int16) - ref["rgb8"].astype(np.int16)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits", "rays"):
        assert out["stats"][key] == ref["counters"][key], key
    assert ref["counters"]["rays"] > 30_000_000


def _workload_text(name):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_scene
    from conftest import scene_path
    if name == "synth10k":
        return gen_scene.synth_spheres(100, 1234, 5, 5)
    if name == "synth90k":
        return gen_scene.synth_mixed(300, 4321, 5, 10)
    return open(scene_path(name)).read()


@pytest.mark.parametrize("name", ["pool", "original", "small", "synth10k", "synth90k"])
def test_scaled_workloads_equal_float64_shadows(name):
    """Every workload uniformly scaled by 0.001 ... 1000 (all lengths: positions, radii, plane offsets, light radii,
    camera): each error bound of the float32 filters has to scale with the scene.  Default path == float64 shadows."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from check_beam_vs_exact import scaled
    from ray_tracing_amd.render import FastScene
    text = _workload_text(name)
    for k in (0.001, 0.01, 0.1, 10.0, 100.0, 1000.0):
        sc = FastScene(text=scaled(text, k))
        with native.Context(sc.packed, device=0) as ctx:
            kw = dict(sh_rays=3, jitter=True, seed=12)
            fast = ctx.render(abi.render_params(640, 360, **kw))
            exact = ctx.render(abi.render_params(640, 360, flags=abi.RT_FLAG_NO_BEAM, **kw))
        assert np.array_equal(fast["ids"], exact["ids"]), (name, k)
        assert int((fast["rgb"] != exact["rgb"]).any(-1).sum()) == 0, (name, k)


def _grazing_scene(extent, seed):
    """Spheres of very different sizes scattered over a field `extent` wide, lights less than one degree above the
    horizon of most shaded points and `extent` away from them: shadow rays skim silhouettes at sample-to-centre distances
    R up to ~extent, where the reference's own float64 rounding of b^2 - 4ac moves entry points by more than the 1e-6
    of its lit test - the band the float32 filters must leave to the float64 queue."""
    rng = np.random.default_rng(seed)
    L = ["cam 0 %r %r 0 0 0 0 1 0 1.2 1" % (0.02 * extent, -0.55 * extent), "set 0.3 0.4 0.6 3 2",
         "mtl 0.8 0.4 0.3 0.6 0.6 0.6 0.2 0.2 0.2 20 0", "mtl 0.3 0.7 0.4 0.3 0.3 0.3 0 0 0 6 0",
         "mtl 0.5 0.5 0.55 0.1 0.1 0.1 0 0 0 2 0", "pln 0 1 0 0 3"]
    n = 14
    for ix in range(n):
        for iz in range(n):
            r = float(extent * 10 ** rng.uniform(-3.3, -1.6))
            x = (ix - n / 2 + 0.5 + rng.uniform(-0.3, 0.3)) * extent / n
            z = (iz - n / 2 + 0.5 + rng.uniform(-0.3, 0.3)) * extent / n
            L.append("sph %r %r %r %r %d" % (x, r * float(rng.uniform(0.2, 1.0)), z, r, 1 + (ix + iz) % 2))
    L.append("lgt %r %r %r 1 1 0.9 1 0.9 %r" % (-0.9 * extent, 0.012 * extent, 0.3 * extent, 0.004 * extent))     # ~0.7 degrees
    L.append("lgt %r %r %r 0.5 0.5 0.7 1 0.7 %r" % (0.7 * extent, 0.004 * extent, -0.8 * extent, 0.001 * extent))  # ~0.2 degrees
    return "\n".join(L) + "\n"


@pytest.mark.parametrize("extent,seed", [(100.0, 1), (1e4, 2), (1e4, 3), (3e5, 4), (2e6, 5), (0.5, 6)])
def test_grazing_silhouettes_under_lights_near_the_horizon(oracle, extent, seed):
    from ray_tracing_amd.render import FastScene
    sc = FastScene(text=_grazing_scene(extent, seed))
    with native.Context(sc.packed, device=0) as ctx:
        kw = dict(jitter=True, seed=5)
        fast = ctx.render(abi.render_params(480, 270, **kw))
        f64 = ctx.render(abi.render_params(480, 270, flags=abi.RT_FLAG_NO_BEAM, **kw))
        exact = ctx.render(abi.render_params(480, 270, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **kw))
        p = abi.render_params(480, 270, rank=1, world=6, **kw)
        part = ctx.render(p)
    assert np.array_equal(fast["ids"], exact["ids"]) and np.array_equal(fast["ids"], f64["ids"])
    assert int((fast["rgb"] != f64["rgb"]).any(-1).sum()) == 0
    assert int((fast["rgb"] != exact["rgb"]).any(-1).sum()) == 0
    lit = fast["rgb"].mean()
    assert 0.02 < lit < 0.98                       # (the frame shows shadows and lit ground, not one flat colour)
    ref = oracle.render(sc.packed, p, threads=min(16, os.cpu_count() or 1))
    assert np.array_equal(part["ids"], ref["ids"])
    assert np.abs(part["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    assert part["stats"]["shadow_rays"] == ref["counters"]["shadow_rays"]


def test_scene_beyond_the_float32_range_of_the_filters_takes_the_float64_paths(oracle):
    """Coordinates of ~1e7 and more: ray_is_tame / the `delicate` test of k_beam_setup send such rays and pairs to the
    reference-order walk and the float64 queue (range of the margins).  Same frame as the exact walk."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from check_beam_vs_exact import scaled
    from conftest import scene_path
    from ray_tracing_amd.render import FastScene
    sc = FastScene(text=scaled(open(scene_path("small")).read(), 3e6))
    with native.Context(sc.packed, device=0) as ctx:
        kw = dict(sh_rays=3, rec_max=3, jitter=True, seed=2)
        fast = ctx.render(abi.render_params(160, 90, **kw))
        exact = ctx.render(abi.render_params(160, 90, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **kw))
    assert np.array_equal(fast["ids"], exact["ids"]) and np.array_equal(fast["rgb"], exact["rgb"])
    ref = oracle.render(sc.packed, abi.render_params(160, 90, **kw), threads=min(16, os.cpu_count() or 1))
    assert np.array_equal(fast["ids"], ref["ids"])
    assert np.abs(fast["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
