"""rt_visibility - the any-hit form of the batched query: the lit mask Light.compute_shadows
takes from every shadow ray (Light.py:214-221).  Bar: bit-exact verdicts, against the reference's own masks
(tests/golden/visibility_*.npz, made by tools/make_golden.py) and against the oracle's nearest hit + the same
comparison on arbitrary batches."""
import numpy as np
import pytest

from conftest import golden, load_scene

pytestmark = pytest.mark.gpu

TAGS = ["original", "small", "pool", "synth400"]


def lit_from_points(points, targets):
    with np.errstate(invalid="ignore"):
        return np.linalg.norm(points - targets, axis=-1) < 1e-6          # Light.py:220-221, NaN -> False


@pytest.mark.parametrize("tag", TAGS)
def test_visibility_matches_reference_masks(gpu_ctx, tag):
    g, meta = golden("visibility_" + tag)
    targets = g["hits"][g["target_index"]]
    lit = gpu_ctx(tag).visibility(g["origins"], g["directions"], targets)
    assert lit.dtype == bool and lit.shape == (meta["rays"],)
    assert np.array_equal(lit, g["lit"].astype(bool))
    # the same verdicts from the nearest-hit operator + the host comparison
    pts, _ = gpu_ctx(tag).intersect(g["origins"], g["directions"])
    assert np.array_equal(lit, lit_from_points(pts, targets))


@pytest.mark.parametrize("tag", TAGS)
def test_visibility_arbitrary_batches_match_oracle(gpu_ctx, oracle, tag):
    """Targets off the ray, targets that are other rays' hit points, non-unit / zero / NaN directions, far origins."""
    sc = load_scene(tag)
    f = sc["bsp"].flat
    rng = np.random.default_rng(21)
    n = 6000
    lo, hi = np.asarray(f.lo[0]), np.asarray(f.hi[0])
    o = rng.uniform(lo - 2, hi + 2, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    pts0, ids0, _ = oracle.intersect(sc["packed"], o, d)
    t = pts0.copy()                                           # the ray's own nearest hit: lit unless it is a miss (NaN)
    t[1::4] = pts0[rng.permutation(n)][1::4]                  # somebody else's hit point
    t[2::4] = o[2::4] + d[2::4] * rng.uniform(0.1, 30, (len(o[2::4]), 1))      # a point on the ray
    t[3::4] = rng.uniform(lo, hi, (len(o[3::4]), 3))          # anywhere
    k = np.arange(n)
    d[k % 10 == 5] *= 0.8                                     # non-unit, still the fast walk
    d[k % 10 == 6] *= 7.0                                     # non-unit, exact walk
    d[k % 50 == 7] = 0.0
    d[k % 50 == 8, 1] = np.nan
    o[k % 50 == 9] *= 1e9
    t[k % 50 == 10] = 1e8                                     # the Background point (Background.py:23)
    t[k % 50 == 11] += 5e-7                                   # inside the EPSILON ball of a hit
    t[k % 50 == 12] += 2e-6                                   # just outside
    pts, _, _ = oracle.intersect(sc["packed"], o, d)
    want = lit_from_points(pts, t)
    got = gpu_ctx(tag).visibility(o, d, t)
    assert np.array_equal(got, want), "%d verdicts differ" % (got != want).sum()
    assert 0 < want.sum() < n


def test_visibility_through_the_host_mirror(gpu_ctx):
    sc = load_scene("small")
    g, _ = golden("visibility_small")
    targets = g["hits"][g["target_index"]]
    bsp = sc["bsp"]
    sel = slice(0, 4000)
    mask = bsp.direct_light_mask(g["origins"][sel], g["directions"][sel], targets[sel])
    pts, _ = bsp.intersect_vectorize(g["origins"][sel], g["directions"][sel])
    assert np.array_equal(mask, lit_from_points(pts, targets[sel]))
    assert np.array_equal(mask, g["lit"][sel].astype(bool))


def test_visibility_argument_errors(gpu_ctx):
    ctx = gpu_ctx("small")
    z = np.zeros((4, 3))
    assert ctx.visibility(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3))).shape == (0,)
    with pytest.raises(ValueError):
        ctx.visibility(z, z, np.zeros((3, 3)))
    with pytest.raises(ValueError):
        ctx.visibility(z.ravel(), z.ravel(), z.ravel())
