import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from ray_tracing_amd import abi, native
for wl, W, H, n in (("original", 500, 500, 80), ("synth1M", 1920, 1080, 30)):
    scene, _ = bench.build_scene(wl)
    with native.Context(scene.packed) as ctx:
        for seed in [1] + list(range(10, 20)) + list(range(100, 100 + n)):
            try:
                out = ctx.render(abi.render_params(W, H, seed=seed), want_rgb=False, want_ids=False)
            except native.NativeError as e:
                print(wl, "seed", seed, "->", e, flush=True)
        print(wl, "done", ctx.frame_bytes() / 1e9, "GB", flush=True)
