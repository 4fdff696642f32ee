#!/usr/bin/env python3
"""Frame time of frames enqueued BACK TO BACK on one stream (no synchronisation between them: bench.time_frames) - what a
renderer that keeps frames in flight sees.  usage: python tools/b2b.py [--workload W] [--width X --height Y] [--world N --rank R]
[--sh-rays S] [--frames K].  RT_AMD_NO_EARLY_CAM=1 for the schedule without the early camera pass."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth10k")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--sh-rays", type=int, default=None)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1)
    a = ap.parse_args()
    import torch
    import bench
    from ray_tracing_amd import abi, native
    scene, _ = bench.build_scene(a.workload)
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    kw = dict(rank=a.rank, world=a.world)
    if a.sh_rays is not None:
        kw["sh_rays"] = a.sh_rays
    with native.Context(scene.packed) as ctx:
        ms = [bench.time_frames(ctx, abi, native, torch, stream, a.width, a.height, a.frames, 10, **kw) for _ in range(3)]
        # host side: how long the calls of a frame take to return (the GPU idle, nothing to wait for)
        import time
        n = native.tile_pixels(abi.render_params(a.width, a.height, **kw))
        buf = torch.empty(n * 3, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for i in range(a.frames):
                ctx.render_device(abi.render_params(a.width, a.height, seed=300 + i, **kw), d_rgb8=buf.data_ptr(), stream=stream.cuda_stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("host: %.3f ms per frame to enqueue, %.3f ms per frame until the last one is done" % ((t1 - t0) * 1e3 / a.frames, (t2 - t0) * 1e3 / a.frames))
        print("b2b %s %dx%d world %d: ms per frame %s | frame state %.2f GB" % (a.workload, a.width, a.height, a.world, [round(m, 3) for m in ms], ctx.frame_bytes() / 1e9))


if __name__ == "__main__":
    main()
