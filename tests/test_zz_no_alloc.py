"""rt_render_device_noalloc (include/rt_amd.h): the frame entry point for callers that cannot allocate.

It has no counterpart in the reference; its contract is "the frame rt_render_device gives, with all working memory in
the library's static device array and no allocating HIP call".  Checked here through the C-ABI, bit for bit (float32
frame, 8-bit frame, primary hit ids, ray counters), on two calls in a row, the second with a smaller frame, for the
beam path (s = 3, hashed jitter), the per-ray float64 path (s = 1) and a scene with reflection + transparency."""
import ctypes as C

import numpy as np
import pytest
import torch  # noqa: F401  (before the native library is loaded: the process must end up with ONE HIP runtime, torch's)

from conftest import load_scene
from ray_tracing_amd import abi, native

pytestmark = pytest.mark.gpu


def _frames(ctx, params, entry):
    """One frame through `entry` ('alloc' | 'noalloc') into torch-owned device buffers -> (rgb, rgb8, ids, stats)."""
    n = native.tile_pixels(params)
    d_rgb = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda:0")
    d_rgb8 = torch.full((n, 3), 77, dtype=torch.uint8, device="cuda:0")
    d_ids = torch.full((n,), -5, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    if entry == "noalloc":
        before = native.lib().rt_debug_alloc_calls()
        ctx.render_device_noalloc(params, d_rgb.data_ptr(), d_rgb8.data_ptr(), d_ids.data_ptr())
        assert native.lib().rt_debug_alloc_calls() == before, "the no-alloc entry point issued an allocating HIP call"
    else:
        ctx.render_device(params, d_rgb.data_ptr(), d_rgb8.data_ptr(), d_ids.data_ptr())
    stats = ctx.sync_stats()
    return d_rgb.cpu().numpy(), d_rgb8.cpu().numpy(), d_ids.cpu().numpy(), stats


COUNTERS = ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits", "levels", "max_queue")


@pytest.mark.parametrize("tag,sh,jitter", [("pool", 3, True), ("pool", 1, False), ("original", 3, True), ("small", 2, True)])
def test_noalloc_frames_equal_allocating_frames(tag, sh, jitter):
    packed = load_scene(tag)["packed"]
    with native.Context(packed, device=0) as ctx:           # a fresh context: the first frame it sees is a no-alloc one
        for (w, h) in ((96, 72), (40, 24)):                  # two calls in a row, the second one smaller
            p = abi.render_params(w, h, sh_rays=sh, jitter=jitter, seed=11)
            got = _frames(ctx, p, "noalloc")
            ref = _frames(ctx, p, "alloc")
            assert np.array_equal(got[2], ref[2]), "primary hit ids differ (%s %dx%d)" % (tag, w, h)
            assert np.array_equal(got[0], ref[0]), "float32 frame differs (%s %dx%d)" % (tag, w, h)
            assert np.array_equal(got[1], ref[1]), "8-bit frame differs (%s %dx%d)" % (tag, w, h)
            for k in COUNTERS:
                assert got[3][k] == ref[3][k], k
        # and once more after the allocating frames, on a tile partition (rank 1 of 3)
        p = abi.render_params(64, 48, sh_rays=sh, jitter=jitter, seed=5, rank=1, world=3)
        got, ref = _frames(ctx, p, "noalloc"), _frames(ctx, p, "alloc")
        for a, b in zip(got[:3], ref[:3]):
            assert np.array_equal(a, b)


def test_noalloc_rejects_what_does_not_fit():
    packed = load_scene("pool")["packed"]
    assert native.lib().rt_noalloc_scratch_bytes() == 64 << 20
    with native.Context(packed, device=0) as ctx:
        p = abi.render_params(1920, 1080, sh_rays=1, jitter=False)
        with pytest.raises(native.NativeError) as e:
            ctx.render_device_noalloc(p)
        assert e.value.code == abi.RT_ERR_UNSUPPORTED and "does not fit" in str(e.value)
        p = abi.render_params(32, 32, sh_rays=1, jitter=False, flags=abi.RT_FLAG_PROFILE)
        with pytest.raises(native.NativeError) as e:
            ctx.render_device_noalloc(p)
        assert e.value.code == abi.RT_ERR_UNSUPPORTED
        # the context is still good for both entry points
        p = abi.render_params(32, 32, sh_rays=2, jitter=True, seed=1)
        a, b = _frames(ctx, p, "noalloc"), _frames(ctx, p, "alloc")
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])


def test_noalloc_recursion_below_zero_is_background():
    packed = load_scene("pool")["packed"]
    with native.Context(packed, device=0) as ctx:
        p = abi.render_params(24, 16, sh_rays=1, jitter=False, rec_max=-1)
        a, b = _frames(ctx, p, "noalloc"), _frames(ctx, p, "alloc")
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        bg = np.clip(np.asarray(packed.desc.background[:], dtype=np.float64), 0, 1).astype(np.float32)
        assert np.array_equal(a[0], np.broadcast_to(bg, a[0].shape))
