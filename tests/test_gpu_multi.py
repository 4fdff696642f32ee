"""The multi-GPU path on ONE GPU at BASELINE.json configs[3]'s size: the 8 ranks' shares of the 3840x2160 frame of the
10 k-sphere scene are rendered one after the other into one packed device buffer (what RCCL's gather delivers to rank
0), rt_deinterleave_u8 scatters them, and the image must equal the whole-frame render bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before the native library: one HIP runtime per process, torch's)

from ray_tracing_amd import abi, native

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench_ctx():
    sys.path.insert(0, ROOT)
    import bench
    scene, _ = bench.build_scene("synth10k")
    ctx = native.Context(scene.packed, device=0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("w,h,world,tile", [(3840, 2160, 8, 0), (1918, 1078, 3, 0), (640, 360, 5, 16)])
def test_ranks_gathered_and_deinterleaved_on_the_device_equal_the_whole_frame(bench_ctx, w, h, world, tile):
    ctx = bench_ctx
    kw = dict(sh_rays=5, rec_max=5, jitter=True, seed=2024, tile_w=tile, tile_h=tile)
    whole = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda:0")
    ctx.render_device(abi.render_params(w, h, **kw), d_rgb8=whole.data_ptr())
    whole_stats = ctx.sync_stats()
    n = native.tile_pixels(abi.render_params(w, h, rank=0, world=world, **kw))
    packed = torch.full((world, n * 3), 201, dtype=torch.uint8, device="cuda:0")
    rays = 0
    for r in range(world):
        p = abi.render_params(w, h, rank=r, world=world, **kw)
        assert native.tile_pixels(p) == n                      # equal-sized buffers: what a gather needs
        ctx.render_device(p, d_rgb8=packed[r].data_ptr())
        rays += ctx.sync_stats()["rays"]
    image = torch.full((w * h * 3,), 7, dtype=torch.uint8, device="cuda:0")
    ctx.deinterleave_u8(abi.render_params(w, h, rank=0, world=world, **kw), packed.data_ptr(), n * 3, image.data_ptr())
    ctx.sync_stats()                                           # waits for the context's stream
    torch.cuda.synchronize()
    assert rays == whole_stats["rays"]
    assert torch.equal(image, whole)
    if (w, h) == (3840, 2160):
        assert whole_stats["rays"] > 1_000_000_000
    # the NumPy de-interleave of the gloo tests gives the same image
    if w * h <= 1 << 20:
        from ray_tracing_amd.render import deinterleave_host
        host = deinterleave_host(list(packed.cpu().numpy()), w, h, world, tile_w=tile, tile_h=tile)
        assert np.array_equal(host.reshape(-1), whole.cpu().numpy())


def test_overlapped_schedule_at_the_largest_and_smallest_frames(bench_ctx):
    """The three-stream schedule of a frame (level 0's soft shadows | nearest hits of levels 1.. | their shading) against
    the levels in sequence, at 3840x2160 and at a single tile, alternating (events and buffer sets are reused from frame to
    frame): same image, same counters."""
    ctx = bench_ctx
    for (w, h) in ((3840, 2160), (8, 8), (3840, 2160), (1, 1)):
        kw = dict(sh_rays=5, rec_max=5, jitter=True, seed=31)
        n = w * h
        a = torch.zeros(n * 3, dtype=torch.float32, device="cuda:0")
        b = torch.zeros(n * 3, dtype=torch.float32, device="cuda:0")
        ctx.render_device(abi.render_params(w, h, **kw), d_rgb=a.data_ptr())
        sa = ctx.sync_stats()
        ctx.render_device(abi.render_params(w, h, flags=abi.RT_FLAG_NO_OVERLAP, **kw), d_rgb=b.data_ptr())
        sb = ctx.sync_stats()
        assert torch.equal(a, b), (w, h)
        for key in ("rays", "hits", "reflect_rays", "levels"):
            assert sa[key] == sb[key], key
    assert ctx.voided_frames() == 0
