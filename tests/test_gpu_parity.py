"""GPU parity tests proper: the HIP path, called through the C-ABI, against the
golden vectors of the reference (tests/golden, made by tools/make_golden.py) and
against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): hit ids bit-exact; deterministic frames within
1/255 per pixel on 8-bit output (we observe 0); float64 hit points bit-exact
where the reference does not go through BLAS."""
import os

import numpy as np
import pytest

from conftest import golden, load_scene
from ray_tracing_amd import abi, native

pytestmark = pytest.mark.gpu


def eq_nan(a, b):
    return np.array_equal(a, b, equal_nan=True)


# ------------------------------------------------------------------ kernel level
def test_unit_primitives_match_reference():
    g, _ = golden("prims")
    o, d = g["origins"], g["directions"]
    from surfaces.Cube import Cube
    from surfaces.InfinitePlane import InfinitePlane
    from surfaces.Sphere import Sphere
    sph = Sphere(g["sphere_c"], float(g["sphere_r"]), 1, 1)
    assert eq_nan(sph.intersect_vectorized(o, d), g["sphere_hits"])
    ms = ~np.isnan(g["sphere_hits"][:, 0])
    assert np.array_equal(sph.calculate_normals(g["sphere_hits"][ms]), g["sphere_normals"][ms])
    for key in ("cube", "cube2"):
        cub = Cube(g[key + "_c"], float(g[key + "_s"]), 1, 2)
        assert eq_nan(cub.intersect_vectorized(o, d), g[key + "_hits"])
    cub = Cube(g["cube_c"], float(g["cube_s"]), 1, 2)
    mc = ~np.isnan(g["cube_hits"][:, 0])
    assert np.array_equal(cub.calculate_normals(g["cube_hits"][mc]), g["cube_normals"][mc])
    assert np.array_equal(cub.calculate_normals(g["cube_pts"]), g["cube_pts_normals"])
    pln = InfinitePlane(g["plane_n_in"], float(g["plane_off_in"]), 1, 4)
    hp = pln.intersect_vectorized(o, d)
    assert np.array_equal(np.isnan(hp), np.isnan(g["plane_hits"]))
    # np.dot goes through BLAS in the reference: relative 1e-12 instead of bit equality
    ok = ~np.isnan(hp)
    assert np.allclose(hp[ok], g["plane_hits"][ok], rtol=1e-12, atol=1e-12)
    m, tn = Cube.cube_intersection_mask(o, d, g["box_lo"], g["box_hi"], outside_hits_only=False)
    assert np.array_equal(m[:, 0], g["node_mask"])
    m, tn = Cube.cube_intersection_mask(o, d, g["box_lo"], g["box_hi"], outside_hits_only=True)
    assert np.array_equal(m[:, 0], g["prim_mask"])
    # np.max/np.min propagate NaN, fmin/fmax do not: t_near is compared where the reference's is a number
    fin = ~np.isnan(g["t_near"])
    assert np.array_equal(tn[fin, 0], g["t_near"][fin])


@pytest.mark.parametrize("tag,w,h", [("pool", 16, 9), ("pool", 7, 5), ("original", 12, 12), ("small", 9, 6)])
def test_initial_rays_bit_exact(gpu_ctx, tag, w, h):
    g, _ = golden("rays_%s_%dx%d" % (tag, w, h))
    src, dirs = gpu_ctx(tag).initial_rays(w, h)
    assert np.array_equal(src, g["sources"])
    assert np.array_equal(dirs, g["directions"])


# ------------------------------------------------------------------ nearest hit
@pytest.mark.parametrize("tag,w,h", [("pool", 1920, 1080), ("original", 500, 500), ("small", 320, 200),
                                     ("synth400", 320, 180)])
def test_primary_hit_ids_bit_exact(gpu_ctx, tag, w, h):
    g, meta = golden("primary_%s_%dx%d" % (tag, w, h))
    ctx = gpu_ctx(tag)
    out = ctx.render(abi.render_params(w, h, sh_rays=1, rec_max=0, jitter=False), want_rgb=False)
    assert np.array_equal(out["ids"], g["ids_q4fix"].astype(np.int32))
    if meta["q4_events"] == 0:
        assert np.array_equal(out["ids"], g["ids"].astype(np.int32))
    # the same through the intersect_vectorize drop-in, points included
    src, dirs = ctx.initial_rays(w, h)
    sel = g["sample_index"]
    pts, ids = ctx.intersect(src[sel], dirs[sel])
    assert np.array_equal(ids, g["ids_q4fix"][sel].astype(np.int64))
    assert np.allclose(pts, g["sample_points"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag", ["original", "small", "pool", "synth400"])
def test_intersect_incoherent_rays(gpu_ctx, oracle, tag):
    g, _ = golden("secondary_" + tag)
    pts, ids = gpu_ctx(tag).intersect(g["origins"], g["directions"])
    assert ids.dtype == np.int64 and np.array_equal(ids, g["ids"].astype(np.int64))
    assert np.allclose(pts, g["points"], rtol=1e-12, atol=1e-12, equal_nan=True)
    # and bit-for-bit against the oracle (same operation order on both sides)
    opts, oids, _ = oracle.intersect(load_scene(tag)["packed"], g["origins"], g["directions"])
    assert np.array_equal(ids, oids)
    assert eq_nan(pts, opts)


def test_rays_in_a_node_plane_are_culled_like_the_reference(gpu_ctx, oracle):
    """Origin exactly on a node bound with direction exactly 0 along that axis: the reference culls the node
    (0 * inf = NaN, Cube.py:120-136).  Golden made by the reference on every node of the mirror-symmetric scene."""
    g, _ = golden("secondary_oncut")
    ctx = gpu_ctx("oncut")
    pts, ids = ctx.intersect(g["origins"], g["directions"])
    assert np.array_equal(ids, g["ids"].astype(np.int64))
    assert np.allclose(pts, g["points"], rtol=1e-12, atol=1e-12, equal_nan=True)
    opts, oids, _ = oracle.intersect(load_scene("oncut")["packed"], g["origins"], g["directions"])
    assert np.array_equal(ids, oids) and eq_nan(pts, opts)
    # the any-hit query on the same rays: lit <=> nearest hit within 1e-6 of the target
    tg = np.where((g["ids"] != 0)[:, None], g["points"], g["origins"] + g["directions"])   # a miss: some point on the ray
    lit = ctx.visibility(g["origins"], g["directions"], tg)
    assert np.array_equal(lit, g["ids"] != 0)
    for w, h in ((65, 39), (9, 5)):
        gp, m = golden("primary_oncut_%dx%d" % (w, h))
        out = ctx.render(abi.render_params(w, h, sh_rays=1, rec_max=0, jitter=False), want_rgb=False)
        assert np.array_equal(out["ids"], gp["ids"].astype(np.int32))
        # hashed jitter on: default path == reference-order walk == per-ray float64 path, bit for bit
        frames = [ctx.render(abi.render_params(w, h, sh_rays=3, jitter=True, seed=9, flags=f))
                  for f in (0, abi.RT_FLAG_EXACT_TRAVERSAL, abi.RT_FLAG_NO_BEAM)]
        for fr in frames[1:]:
            assert np.array_equal(fr["rgb"], frames[0]["rgb"]) and np.array_equal(fr["ids"], frames[0]["ids"])


# ------------------------------------------------------------------ frames, deterministic
FRAMES = ["pool_64x36_s1", "pool_160x90_s5", "pool_1920x1080_s1", "original_100x100", "original_500x500",
          "small_160x100", "small_64x40_s3_r6", "synth400_160x90_s2_r3",
          # camera, a light and the root cut on one plane, odd width: rays with direction.x == 0 from origins on a node bound
          "oncut_65x39_s1", "oncut_65x39_s3", "oncut_43x25_s2_r1"]


@pytest.mark.parametrize("name", FRAMES)
def test_frame_matches_reference(gpu_ctx, name):
    g, meta = golden("frame_" + name + "_nojitter")
    w, h = meta["width"], meta["height"]
    out = gpu_ctx(meta["scene"]).render(abi.render_params(w, h, sh_rays=meta["sh_rays"], rec_max=meta["rec_max"],
                                                          jitter=False))
    img = out["rgb8"].reshape(h, w, 3).astype(int)
    diff = np.abs(img - g["rgb8_q4fix"].astype(int))
    assert diff.max() <= 1, "pixels off by more than 1/255: %d" % (diff.max(-1) > 1).sum()
    if meta["q4_events"] == 0:
        assert np.abs(img - g["rgb8"].astype(int)).max() <= 1
        assert out["stats"]["rays"] == meta["rays"]
    if "rgb_f64_q4fix" in g.files:
        ref = g["rgb_f64_q4fix"].clip(0, 1)
        assert np.abs(out["rgb"].astype(np.float64) - ref).max() < 1e-6


@pytest.mark.parametrize("tag,w,h,sh,rec", [("small", 96, 60, 2, 4), ("original", 64, 64, 2, 5),
                                            ("pool", 128, 72, 3, 10)])
def test_frame_matches_oracle_with_jitter(gpu_ctx, oracle, tag, w, h, sh, rec):
    """Same counter-based jitter hash on both sides -> images agree to the last 8-bit step."""
    p = abi.render_params(w, h, sh_rays=sh, rec_max=rec, jitter=True, seed=1234)
    out = gpu_ctx(tag).render(p)
    ref = oracle.render(load_scene(tag)["packed"], p)
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
        assert out["stats"][key] == ref["counters"][key], key


def test_trace_arbitrary_batch_matches_oracle(gpu_ctx, oracle):
    g, _ = golden("secondary_small")
    p = abi.render_params(1, 1, sh_rays=2, rec_max=3, jitter=False)
    o, d = g["origins"][:1500], g["directions"][:1500]
    rgb, st = gpu_ctx("small").trace(p, o, d)
    ref, cnt = oracle.trace(load_scene("small")["packed"], p, o, d)
    assert np.allclose(rgb, ref, rtol=1e-9, atol=1e-12)
    assert st["rays"] == cnt["rays"]


# ------------------------------------------------------------------ stochastic soft shadows
@pytest.mark.parametrize("name,sh", [("pool_192x108_s5", 5), ("pool_192x108_s2", 2)])
def test_soft_shadow_psnr_vs_reference_mean(gpu_ctx, name, sh):
    """PSNR >= 40 dB between the mean of K GPU frames (hash jitter) and the mean of K seeded
    reference frames at the same s."""
    g, meta = golden("stoch_%s_mean%d" % (name, 32))
    w, h, K = meta["width"], meta["height"], 64
    acc = np.zeros((h * w, 3))
    ctx = gpu_ctx("pool")
    for k in range(K):
        out = ctx.render(abi.render_params(w, h, sh_rays=sh, rec_max=meta["rec_max"], jitter=True, seed=77 + k),
                         want_ids=False)
        acc += out["rgb"]
    mean = (acc / K).reshape(h, w, 3)
    mse = np.mean((mean - g["mean_rgb"].astype(np.float64)) ** 2)
    psnr = 10 * np.log10(1.0 / mse)
    # the reference mean itself carries Monte-Carlo noise of K=32 frames; 40 dB is the north-star bar
    assert psnr >= 40.0, psnr


def test_mean_of_frames_on_the_device_equals_the_mean_of_single_frames(gpu_ctx):
    """rt_render_mean = the frames rt_render gives for seed .. seed+K-1, clipped, averaged (float64 on the device)."""
    w, h, K = 160, 90, 7
    ctx = gpu_ctx("pool")
    kw = dict(sh_rays=3, rec_max=4, jitter=True)
    acc = np.zeros((h * w, 3))
    rays = 0
    for k in range(K):
        one = ctx.render(abi.render_params(w, h, seed=900 + k, **kw), want_ids=False)
        acc += one["rgb"].astype(np.float64)
        rays += one["stats"]["rays"]
    out = ctx.render_mean(abi.render_params(w, h, seed=900, **kw), K)
    assert np.abs(out["rgb"].astype(np.float64) - acc / K).max() < 1e-6      # float32 outputs on both sides
    assert np.abs(out["rgb8"].astype(int) - np.floor(acc / K * 255).astype(int)).max() <= 1
    assert out["stats"]["rays"] == rays
    # K = 1 is rt_render itself, bit for bit; a rank's tiles average like the whole frame
    single = ctx.render(abi.render_params(w, h, seed=900, **kw), want_ids=False)
    m1 = ctx.render_mean(abi.render_params(w, h, seed=900, **kw), 1)
    assert np.array_equal(m1["rgb"], single["rgb"]) and np.array_equal(m1["rgb8"], single["rgb8"])
    p = abi.render_params(w, h, seed=900, rank=1, world=3, **kw)
    part = ctx.render_mean(p, K)
    pix = native.tile_layout(p)
    ok = pix >= 0
    assert np.array_equal(part["rgb8"][ok], out["rgb8"][pix[ok]])
    with pytest.raises(native.NativeError):
        ctx.render_mean(abi.render_params(w, h, **kw), 0)


def test_mean_mode_psnr_vs_reference_mean(gpu_ctx):
    g, meta = golden("stoch_pool_192x108_s2_mean32")
    w, h = meta["width"], meta["height"]
    out = gpu_ctx("pool").render_mean(abi.render_params(w, h, sh_rays=2, rec_max=meta["rec_max"], jitter=True, seed=77), 64)
    mse = np.mean((out["rgb"].reshape(h, w, 3).astype(np.float64) - g["mean_rgb"].astype(np.float64)) ** 2)
    assert 10 * np.log10(1.0 / mse) >= 40.0


# ------------------------------------------------------------------ tiles / multi-GPU layout on one GPU
def test_tile_partition_reassembles_bit_identical(gpu_ctx):
    w, h, world = 100, 52, 4
    ctx = gpu_ctx("small")
    full = ctx.render(abi.render_params(w, h, jitter=True, seed=5))
    img = np.zeros((h * w, 3), dtype=np.uint8)
    ids = np.zeros(h * w, dtype=np.int32)
    for r in range(world):
        p = abi.render_params(w, h, jitter=True, seed=5, rank=r, world=world)
        part = ctx.render(p)
        pix = native.tile_layout(p)
        ok = pix >= 0
        img[pix[ok]] = part["rgb8"][ok]
        ids[pix[ok]] = part["ids"][ok]
    assert np.array_equal(ids, full["ids"])
    assert np.array_equal(img, full["rgb8"])


def test_large_shadow_grids_grow_the_float64_queue(gpu_ctx):
    """16 x 16 shadow rays on the box-crowded scene: most pairs leave the filter undecided, 256 rays each - the float64
    queue has to grow on its own (the ray queues do not), and the frame equals the exact walk."""
    from ray_tracing_amd import native as nat
    ctx = nat.Context(load_scene("original")["packed"], device=0)      # a fresh context: queue sizes at their defaults
    try:
        p = dict(sh_rays=16, rec_max=3, jitter=True, seed=9)
        fast = ctx.render(abi.render_params(96, 96, **p))
        exact = ctx.render(abi.render_params(96, 96, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **p))
    finally:
        ctx.close()
    assert fast["stats"]["shadow_rays"] > 96 * 96 * 256
    assert np.array_equal(fast["ids"], exact["ids"])
    assert np.array_equal(fast["rgb"], exact["rgb"])


# ------------------------------------------------------------------ error behaviour of the boundary
def test_bad_arguments_are_reported(gpu_ctx):
    ctx = gpu_ctx("pool")
    with pytest.raises(native.NativeError):
        ctx.render(abi.render_params(0, 10))
    with pytest.raises(native.NativeError):
        ctx.render(abi.render_params(8, 8, sh_rays=0))
    with pytest.raises(ValueError):
        ctx.intersect(np.zeros((4, 3)), np.zeros((5, 3)))


# ------------------------------------------------------------------ fast path == exact walk
@pytest.mark.parametrize("tag,w,h,sh,rec", [("small", 160, 100, 3, 5), ("original", 200, 200, 2, 10),
                                            ("pool", 320, 180, 5, 10), ("synth400", 320, 180, 5, 5),
                                            ("pool", 640, 360, 1, 10), ("small", 320, 200, 1, 2)])
def test_fast_traversal_equals_reference_order_walk(gpu_ctx, tag, w, h, sh, rec):
    """float32-culled kd traversal (default) against the float64 reference-order walk
    (RT_FLAG_EXACT_TRAVERSAL): ids, ray counts and the float32 image must be identical."""
    ctx = gpu_ctx(tag)
    exact = ctx.render(abi.render_params(w, h, sh_rays=sh, rec_max=rec, jitter=True, seed=42,
                                         flags=abi.RT_FLAG_EXACT_TRAVERSAL))
    # default = beam + float32 filtered shadow predicates; NO_BEAM = one float64-confirmed traversal per shadow ray
    for flags in (0, abi.RT_FLAG_NO_BEAM):
        fast = ctx.render(abi.render_params(w, h, sh_rays=sh, rec_max=rec, jitter=True, seed=42, flags=flags))
        assert np.array_equal(fast["ids"], exact["ids"])
        assert fast["stats"]["rays"] == exact["stats"]["rays"]
        assert np.array_equal(fast["rgb8"], exact["rgb8"])
        assert np.array_equal(fast["rgb"], exact["rgb"])      # same lit counts -> same float64 sums


def test_fast_intersect_axis_parallel_and_degenerate_rays(gpu_ctx, oracle):
    """Rays with zero direction components, origins on split planes and inside primitives."""
    sc = load_scene("small")
    f = sc["bsp"].flat
    rng = np.random.default_rng(1)
    n = 3000
    o = rng.uniform(f.lo[0] - 1, f.hi[0] + 1, (n, 3))
    d = np.zeros((n, 3))
    d[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)      # axis parallel
    d[1::3] += rng.normal(size=(len(d[1::3]), 3)) * 1e-3                        # nearly axis parallel
    cuts = f.hi[1:40, :]                                                        # origins exactly on node planes,
    o[:117:3] = cuts.reshape(-1, 3)[:39]                                        # directions with two exact zeros
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    pts, ids = gpu_ctx("small").intersect(o, d)
    opts, oids, _ = oracle.intersect(sc["packed"], o, d)
    assert np.array_equal(ids, oids)
    assert np.array_equal(pts, opts, equal_nan=True)


# ------------------------------------------------------------------ whole-pair shadow verdicts under stress
def _stress_scene(kind):
    """Sphere fields with lights chosen to exercise every branch of the beam path: wide penumbrae (large light
    close to the field), hard shadows (umbra verdicts), a light inside the field (cone test disabled, samples
    inside primitives), tiny and huge spheres, terminators on large spheres, boxes mixed in."""
    rng = np.random.default_rng(11)
    L = ["cam 0 6 -14 0 0.5 0 0 1 0 1.3 1", "set 0.2 0.3 0.5 5 3",
         "mtl 0.8 0.3 0.3 0.9 0.9 0.9 0.2 0.2 0.2 30 0", "mtl 0.3 0.8 0.3 0.4 0.4 0.4 0 0 0 8 0",
         "mtl 0.5 0.5 0.6 0.1 0.1 0.1 0 0 0 2 0", "mtl 0.4 0.4 0.9 0.7 0.7 0.7 0 0 0 12 0.5",
         "pln 0 1 0 -1 3"]
    for ix in range(12):
        for iz in range(12):
            c = [ix - 5.5 + rng.uniform(-0.2, 0.2), -1 + rng.uniform(0.2, 1.6), iz - 5.5 + rng.uniform(-0.2, 0.2)]
            r = float(rng.uniform(0.05, 0.45))
            if kind == "boxes" and (ix + iz) % 3 == 0:
                L.append("box %r %r %r %r %d" % (c[0], c[1], c[2], 2 * r, 1 + (ix + iz) % 4))
            else:
                L.append("sph %r %r %r %r %d" % (c[0], c[1], c[2], r, 1 + (ix * 7 + iz) % 4))
    L.append("sph 0 2.5 3 2.2 1")                                     # huge sphere: long terminator, big umbra
    L.append("sph 2 -0.97 -3 0.03 2")                                 # tiny sphere touching the ground
    if kind == "wide":
        L += ["lgt 0 7 -2 1 1 1 1 0.9 6", "lgt -6 3 -6 0.6 0.6 0.8 1 0.7 3"]
    elif kind == "hard":
        L += ["lgt 30 60 -40 1 1 1 1 1 0.05", "lgt -40 25 10 0.5 0.5 0.5 1 0.8 0.3"]
    elif kind == "inside":
        L += ["lgt 0.3 0.4 0.2 1 1 1 1 0.9 1.5", "lgt 0 12 0 0.5 0.5 0.5 1 0.5 2", "lgt 3 -0.9 -2 0.4 0.4 0.4 1 0.8 0.4"]
    else:
        L += ["lgt 2 9 -5 1 1 1 1 0.9 2", "lgt -7 5 3 0.6 0.6 0.6 1 0.6 1"]
    return "\n".join(L) + "\n"


_STRESS = {}


def _stress_ctx(kind):
    if kind not in _STRESS:
        import tempfile
        from ray_tracing_amd.render import Scene
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
            f.write(_stress_scene(kind))
        try:
            sc = Scene.from_file(f.name)
        finally:
            os.unlink(f.name)
        _STRESS[kind] = (sc, native.Context(sc.packed, device=0))
    return _STRESS[kind][1]


class _ctx_with_env:
    """A fresh context created under the given RT_AMD_* switches (the library reads them once, in rt_create)."""

    def __init__(self, packed, env):
        self.packed, self.env = packed, env

    def __enter__(self):
        old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        try:
            self.ctx = native.Context(self.packed, device=0)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return self.ctx

    def __exit__(self, *exc):
        self.ctx.close()
        return False


@pytest.mark.parametrize("kind,sh,jitter", [("wide", 5, True), ("wide", 3, False), ("hard", 5, True), ("hard", 16, True),
                                            ("inside", 4, True), ("inside", 7, False), ("boxes", 5, True),
                                            ("boxes", 3, True), ("plain", 9, True)])
def test_beam_path_equals_exact_walk_under_stress(kind, sh, jitter):
    """Whole-pair verdicts (umbra, own sphere at the terminator, quad miss test), the per-ray float32 predicates and
    the float64 queue together must give the frame of the exact walk, bit for bit (float64 sums -> float32)."""
    ctx = _stress_ctx(kind)
    w, h = (160, 90) if sh >= 9 else (256, 144)
    common = dict(sh_rays=sh, rec_max=3, jitter=jitter, seed=7)
    exact = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **common))
    fast = ctx.render(abi.render_params(w, h, **common))
    assert np.array_equal(fast["ids"], exact["ids"])
    assert fast["stats"]["rays"] == exact["stats"]["rays"]
    assert np.array_equal(fast["rgb"], exact["rgb"])


def test_beam_path_frame_matches_oracle_on_stress_scene(oracle):
    ctx = _stress_ctx("wide")
    p = abi.render_params(128, 72, sh_rays=5, rec_max=2, jitter=True, seed=3)
    out = ctx.render(p)
    ref = oracle.render(_STRESS["wide"][0].packed, p)
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    assert out["stats"]["shadow_rays"] == ref["counters"]["shadow_rays"]


@pytest.mark.parametrize("tag,w,h,sh,rec", [("synth400", 320, 180, 5, 5), ("pool", 320, 180, 3, 10), ("small", 160, 100, 2, 4),
                                            ("small", 1, 1, 3, 4), ("pool", 8, 8, 5, 10), ("original", 3, 2, 3, 10)])
def test_overlapped_levels_equal_sequential_levels(gpu_ctx, tag, w, h, sh, rec):
    """The default schedule runs levels 1.. beside the soft shadows of level 0 (second stream, second buffer set);
    RT_FLAG_NO_OVERLAP runs the levels one after the other.  Same frame, same counters, several frames in a row."""
    ctx = gpu_ctx(tag)
    for seed in (1, 2, 3):
        seq = ctx.render(abi.render_params(w, h, sh_rays=sh, rec_max=rec, jitter=True, seed=seed, flags=abi.RT_FLAG_NO_OVERLAP))
        ovl = ctx.render(abi.render_params(w, h, sh_rays=sh, rec_max=rec, jitter=True, seed=seed))
        assert np.array_equal(ovl["ids"], seq["ids"])
        assert np.array_equal(ovl["rgb"], seq["rgb"])
        for key in ("rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
            assert ovl["stats"][key] == seq["stats"][key], key


@pytest.mark.parametrize("env", [{"RT_AMD_SETS": "1"}, {"RT_AMD_SETS": "2"}, {"RT_AMD_NO_PIPELINE": "1"}])
def test_schedules_with_fewer_buffer_sets_give_the_same_frames(gpu_ctx, env):
    """Three buffer sets: levels 1.. pipelined over two streams; two: levels 1.. on one stream beside level 0; one: all
    levels in sequence.  The frame does not depend on it (float32 outputs equal, counters equal)."""
    w, h = 192, 108
    kw = dict(sh_rays=3, rec_max=6, jitter=True)
    ref = [gpu_ctx("small").render(abi.render_params(w, h, seed=s, **kw)) for s in (11, 12)]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with native.Context(load_scene("small")["packed"]) as ctx:
            got = [ctx.render(abi.render_params(w, h, seed=s, **kw)) for s in (11, 12)]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for a, b in zip(got, ref):
        assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["rgb"], b["rgb"])
        assert a["stats"]["rays"] == b["stats"]["rays"] and a["stats"]["levels"] == b["stats"]["levels"]


@pytest.mark.parametrize("kind,sh,jitter", [("hard", 5, True), ("hard", 3, False), ("boxes", 5, True), ("boxes", 3, False),
                                            ("inside", 4, True), ("original", 3, True), ("original", 4, False), ("wide", 5, True),
                                            ("plain", 5, True), ("synth400", 5, True)])
def test_rays_of_overflowing_pairs_walk_or_expand_to_the_same_frame(gpu_ctx, kind, sh, jitter):
    """Pairs whose beam meets more primitives than a pair record holds: their rays either walk the tree themselves with
    the float32 filtered predicates (k_shadow_walk, RT_AMD_WALK_MIN=1 forces it) or are expanded into the float64
    kernel (RT_AMD_WALK_MIN=-1).  Same frame bit for bit, and the same as the reference-order walk."""
    ctx = gpu_ctx(kind) if kind in ("original", "synth400") else _stress_ctx(kind)       # original.txt: every pair overflows
    packed = load_scene(kind)["packed"] if kind in ("original", "synth400") else _STRESS[kind][0].packed
    prm = abi.render_params(160, 100, sh_rays=sh, rec_max=3, jitter=jitter, seed=9)
    exact = ctx.render(abi.render_params(160, 100, sh_rays=sh, rec_max=3, jitter=jitter, seed=9, flags=abi.RT_FLAG_EXACT_TRAVERSAL))
    outs = {}
    for v, early in (("1", "0"), ("-1", "0"), ("1", "1"), ("-1", "1")):      # early: umbra test inside the beam traversal
        with _ctx_with_env(packed, {"RT_AMD_WALK_MIN": v, "RT_AMD_BEAM_EARLY": early}) as c2:      # (read once, in rt_create)
            outs[v + early] = c2.render(prm)
    for out in outs.values():
        assert np.array_equal(out["ids"], exact["ids"]) and np.array_equal(out["rgb"], exact["rgb"])
        assert out["stats"]["rays"] == exact["stats"]["rays"]


TILE_ENV = {"RT_AMD_TILES_MIN": "1", "RT_AMD_TILES_MIN_PRIMS": "1"}      # tile beams on frames / scenes of any size


@pytest.mark.parametrize("kind,sh,jitter", [("synth400", 5, True), ("synth400", 3, False), ("small", 4, True), ("pool", 3, True),
                                            ("original", 3, True), ("hard", 5, True), ("boxes", 5, True), ("boxes", 3, False),
                                            ("inside", 4, True), ("wide", 5, True), ("plain", 6, True)])
def test_tile_beams_give_the_same_frames(gpu_ctx, kind, sh, jitter):
    """Level 0 of large frames: the candidates of a (hit, light) pair come from the row that the binning kernels made for its
    8 x 8 pixel tile (k_block_box, k_bin_zone, k_bin_tiles) instead of the pair's own walk through the tree.  Same frame bit
    for bit as with RT_FLAG_NO_TILES and as the reference-order walk: whole frames with ragged edges, tile partitions,
    frames in a row of different sizes on one context."""
    named = kind in ("synth400", "small", "pool", "original")
    if not named:
        _stress_ctx(kind)
    packed = load_scene(kind)["packed"] if named else _STRESS[kind][0].packed
    with _ctx_with_env(packed, TILE_ENV) as ctx:
        used = 0
        for (w, h, rank, world) in ((200, 120, 0, 1), (67, 45, 0, 1), (256, 144, 1, 3), (8, 8, 0, 1), (131, 77, 1, 2)):
            kw = dict(sh_rays=sh, rec_max=2, jitter=jitter, seed=5, rank=rank, world=world)
            before = native.lib().rt_debug_tile_frames(ctx._h)
            tiles = ctx.render(abi.render_params(w, h, **kw))
            used += native.lib().rt_debug_tile_frames(ctx._h) - before
            plain = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_NO_TILES, **kw))
            exact = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **kw))
            for other in (plain, exact):
                assert np.array_equal(tiles["ids"], other["ids"])
                assert np.array_equal(tiles["rgb"], other["rgb"])
                assert tiles["stats"]["rays"] == other["stats"]["rays"]
        assert used == 5, "the frames did not take the tile beams"


def test_tile_beams_are_a_rule_of_scene_and_frame_size(gpu_ctx):
    """Default thresholds: frames below 1.5 M camera rays (400 k behind a pending frame) and scenes below 256 primitives keep the pairs' own walks."""
    ctx = gpu_ctx("synth400")
    before = native.lib().rt_debug_tile_frames(ctx._h)
    ctx.render(abi.render_params(320, 180, sh_rays=3, rec_max=1, jitter=True, seed=1))
    assert native.lib().rt_debug_tile_frames(ctx._h) == before


def _glass_scene(tmp_path):
    """Every hit in this scene spawns a through-ray AND a mirror ray, so level L holds up to 2^L rays per pixel."""
    from ray_tracing_amd import scene_pack
    from Parser import parse_scene_file
    from BSPNode import BSPNode
    text = ["cam 0 1 -6  0 0.8 0  0 1 0  1.2 1", "set 0.2 0.3 0.5 2 3",
            "mtl 0.6 0.6 0.6  0.5 0.5 0.5  0.5 0.5 0.5  20 0.5",
            "pln 0 1 0 -1 1", "pln 0 0 -1 -6 1", "pln 1 0 0 -5 1", "pln -1 0 0 -5 1"]
    for k in range(12):
        text.append("sph %g %g %g 0.9 1" % (-3.3 + 2.2 * (k % 4), 0.2 + 1.9 * (k // 4), 1.0 + 0.4 * (k % 3)))
    text.append("lgt 0 6 -4  1 1 1  1 0.8 1")
    p = tmp_path / "glass.txt"
    p.write_text("\n".join(text) + "\n")
    cam, st, _o, surfaces, mats, lights = parse_scene_file(str(p))
    return scene_pack.pack_scene(surfaces, mats, lights, cam, st, BSPNode.build_bsp_tree(surfaces))


def test_queue_overflow_grows_the_queues_and_renders_again(oracle, tmp_path):
    """The queues (2 x the primary slots to begin with) overflow, rt_sync_stats reports it, the growth factor doubles and
    rt_render renders again until the frame fits.  The frame that comes out equals the oracle's."""
    packed = _glass_scene(tmp_path)
    prm = abi.render_params(96, 54, jitter=True, seed=3)
    ref = oracle.render(packed, prm)
    with native.Context(packed) as ctx:
        first = ctx.render(prm)
        again = ctx.render(prm)                  # queues are large enough now: one pass
    assert first["stats"]["max_queue"] > 2 * 96 * 54          # more rays on one level than the initial queues held
    for out in (first, again):
        assert np.array_equal(out["ids"], ref["ids"])
        assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
        for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
            assert out["stats"][key] == ref["counters"][key], key


def test_a_voided_frame_leaves_nothing_behind_for_a_smaller_one(oracle, tmp_path):
    """A frame voided by a queue overflow returns early from its shading kernels; the lit counters it had already
    filled must not leak into the next frame when that one is small enough to reuse the buffers as they are."""
    packed = _glass_scene(tmp_path)
    with native.Context(packed) as ctx:
        big = abi.render_params(192, 108, jitter=True, seed=3)
        ctx.render_device(big)
        with pytest.raises(native.NativeError) as e:
            ctx.sync_stats()
        assert e.value.code == abi.RT_ERR_QUEUE_OVERFLOW
        small = abi.render_params(48, 27, jitter=True, seed=3)      # (2 x growth) x 1/16 of the slots: no reallocation
        out = ctx.render(small)
    ref = oracle.render(packed, small)
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    assert out["stats"]["shadow_rays"] == ref["counters"]["shadow_rays"]


def test_mixed_lattice_with_transparency_equals_exact_walk(oracle):
    """BASELINE.json configs[4] in small: spheres and boxes alternating on a lattice, 25 % reflective, 15 % transparent,
    recursion 10 (tools/gen_scene.synth_mixed, grid 40 = 1600 primitives).  Default path == exact walk bit for bit,
    and == the oracle on the 8-bit frame."""
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_scene
    from ray_tracing_amd.render import Scene
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(gen_scene.synth_mixed(40, 4321, 5, 10))
    try:
        sc = Scene.from_file(f.name)
    finally:
        os.unlink(f.name)
    with native.Context(sc.packed, device=0) as ctx:
        for sh, rec, w, h in ((5, 10, 240, 135), (3, 4, 320, 180)):
            common = dict(sh_rays=sh, rec_max=rec, jitter=True, seed=21)
            exact = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **common))
            fast = ctx.render(abi.render_params(w, h, **common))
            assert np.array_equal(fast["ids"], exact["ids"])
            assert fast["stats"]["rays"] == exact["stats"]["rays"]
            assert np.array_equal(fast["rgb"], exact["rgb"])
        p = abi.render_params(96, 54, sh_rays=3, rec_max=6, jitter=True, seed=5)
        out = ctx.render(p)
        ref = oracle.render(sc.packed, p)
        assert np.array_equal(out["ids"], ref["ids"])
        assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
        assert out["stats"]["rays"] == ref["counters"]["rays"]


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_files_render_like_the_oracle(oracle, seed, tmp_path):
    """Random scene files (tests/test_scene_native.py: several planes, lights between the surfaces so that ids and
    list positions part ways (Q8), shadow intensities down to 0, cameras looking anywhere): text -> native tables,
    host tree, then GPU frame == oracle frame (ids, 8-bit colours within 1, ray counters)."""
    from test_scene_native import _random_scene_text, python_path
    p = tmp_path / "scene.txt"
    p.write_bytes(_random_scene_text(100 + seed).encode())
    py = python_path(str(p))
    ns = native.NativeScene(path=str(p))
    ns.set_tree(py["bsp"].flat)
    if ns.info["has_settings"] == 0:
        prm = abi.render_params(64, 48, sh_rays=2 + seed % 3, rec_max=3, jitter=True, seed=seed)
    else:
        prm = abi.render_params(64, 48, rec_max=min(4, int(ns.desc.rec_max)), jitter=True, seed=seed)
    ref = oracle.render(ns, prm)
    with native.Context(ns) as ctx:
        out = ctx.render(prm)
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
        assert out["stats"][key] == ref["counters"][key], key


def test_path_codes_beyond_recursion_level_31(oracle, tmp_path):
    """The jitter hash is keyed by the ray's path code, whose bit string fills 32 bits at recursion level 31 (rt_device.hpp,
    child_path): a mirror floor under a mirror ceiling carries rays through all 41 levels of rec_max = 40; past level 31 the code
    is a hash chain.
    GPU frame == oracle frame with the hashed jitter on (ids, 8-bit colours, every ray counter), and the deep levels are
    really reached."""
    from ray_tracing_amd.render import Scene
    text = "\n".join([
        "cam 0 0.5 -6   0 0.45 0   0 1 0   1.0 1.0",
        "set 0.2 0.3 0.4  3 40",
        "mtl 0.1 0.1 0.1  0.3 0.3 0.3  0.95 0.95 0.95  20 0",      # mirrors
        "mtl 0.6 0.2 0.2  0.5 0.5 0.5  0 0 0  8 0",
        "pln 0 1 0  -0.5  1",                                         # a mirror floor ...
        "pln 0 -1 0  -1.5  1",                                        # ... under a mirror ceiling (planes are one-sided: each faces the other)
        "sph 0 0.5 0  0.6  1",
        "sph 1.6 0.0 -1.0  0.5  2",
        "lgt 3 1.2 -4   1 1 1   1 0.8 0.5",
        "lgt -4 1.0 -3  0.8 0.8 1  1 0.6 0.4",
    ]) + "\n"
    p = tmp_path / "mirrors.txt"
    p.write_text(text)
    sc = Scene.from_file(str(p))
    prm = abi.render_params(48, 32, jitter=True, seed=9)
    ref = oracle.render(sc.packed, prm)
    with native.Context(sc.packed, device=0) as ctx:
        out = ctx.render(prm)
    assert out["stats"]["levels"] >= 35, out["stats"]["levels"]
    assert np.array_equal(out["ids"], ref["ids"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    for key in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"):
        assert out["stats"][key] == ref["counters"][key], key
