"""Full-resolution frames of the three synthetic workloads: the default path (beam verdicts, float32 filtered predicates)
against RT_FLAG_NO_BEAM (every shadow ray decided by the float64 walk) - the float32 images must be identical.
    python tools/check_beam_vs_exact.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from ray_tracing_amd import abi, native
for wl in ("synth1M", "synth90k", "synth10k"):
    scene, _ = bench.build_scene(wl)
    ctx = native.Context(scene.packed, device=0)
    for seed in (5, 6):
        kw = dict(jitter=True, seed=seed)
        fast = ctx.render(abi.render_params(1920, 1080, **kw))
        nobeam = ctx.render(abi.render_params(1920, 1080, flags=abi.RT_FLAG_NO_BEAM, **kw))
        d = np.abs(fast["rgb"].astype(np.float64) - nobeam["rgb"].astype(np.float64)).max(-1)
        bad = np.nonzero(d > 0)[0]
        print(wl, "1920x1080 seed", seed, "pixels differ:", len(bad), bad[:8], "ms", round(fast["stats"]["ms_total"], 2), round(nobeam["stats"]["ms_total"], 2), flush=True)
    ctx.close()
