#!/usr/bin/env python3
"""Throughput of a stream of frames with K contexts in flight (frame i on context i % K, each context on its own stream):
does a second frame fill the GPU while the first one walks its latency chain of deeper levels?
    python tools/frames_in_flight.py [--contexts 2] [--width 3840 --height 2160 --world 8 --rank 0] [--frames 200]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synth10k")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1)
    a = ap.parse_args()
    import torch
    import bench
    from ray_tracing_amd import abi, native
    scene, _ = bench.build_scene(a.workload)
    dev = torch.device("cuda:0")
    for K in sorted({1, a.contexts}):
        ctxs = [native.Context(scene.packed, device=0) for _ in range(K)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        n = native.tile_pixels(abi.render_params(a.width, a.height, rank=a.rank, world=a.world))
        bufs = [torch.empty(n * 3, dtype=torch.uint8, device=dev) for _ in range(K)]

        def frame(i):
            k = i % K
            p = abi.render_params(a.width, a.height, rank=a.rank, world=a.world, jitter=True, seed=i)
            with torch.cuda.stream(streams[k]):
                ctxs[k].render_device(p, d_rgb8=bufs[k].data_ptr(), stream=streams[k].cuda_stream)

        for i in range(4 * K):          # queues settle (a first frame may be voided and its queues grown)
            try:
                frame(i)
                ctxs[i % K].sync_stats()
            except native.NativeError as e:
                if e.code != abi.RT_ERR_QUEUE_OVERFLOW:
                    raise
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.frames):
            frame(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        voided = sum(c.voided_frames(reset=True) for c in ctxs)
        print("%d context(s): %.3f ms per frame over %d frames (voided %d)" % (K, dt / a.frames * 1e3, a.frames, voided))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
