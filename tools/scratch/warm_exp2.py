"""k_intersect_only on the camera rays of original.txt 500x500 and on their reflections, vs k_intersect in the frame."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench
from ray_tracing_amd import abi, native
scene, _ = bench.build_scene("original")
with native.Context(scene.packed) as ctx:
    o, d = ctx.initial_rays(500, 500)
    def run(o, d, tag):
        n = o.shape[0]
        do, dd = torch.tensor(o, device="cuda"), torch.tensor(d, device="cuda")
        pts = torch.empty((n, 3), dtype=torch.float64, device="cuda"); ids = torch.empty(n, dtype=torch.int64, device="cuda")
        ts = torch.cuda.Stream(); st = ts.cuda_stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record(ts)
        for k in range(5):
            ctx.intersect_device(do.data_ptr(), dd.data_ptr(), n, pts.data_ptr(), ids.data_ptr(), stream=st)
            ev[k + 1].record(ts)
        torch.cuda.synchronize()
        print(tag, "n", n, "us:", [round(ev[k].elapsed_time(ev[k + 1]) * 1e3, 1) for k in range(5)], "hits", int((ids != 0).sum()))
        return pts.cpu().numpy(), ids.cpu().numpy()
    p, i = run(o, d, "camera rays")
    # crude secondary rays: from the hit points, mirrored about +y (floors / box tops) with a small offset, hits only
    m = i != 0
    d2 = d[m].copy(); d2[:, 1] = -d2[:, 1]
    o2 = p[m] + 1e-6 * d2
    p2, i2 = run(o2, d2, "mirrored")
    m2 = i2 != 0
    d3 = d2[m2].copy(); d3[:, 0] = -d3[:, 0]
    run(p2[m2] + 1e-6 * d3, d3, "mirrored twice")
    sub = np.random.default_rng(1).choice(o2.shape[0], 25000, replace=False)
    run(o2[sub], d2[sub], "25k of the mirrored")
