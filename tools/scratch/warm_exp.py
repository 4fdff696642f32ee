"""Same nearest-hit kernel launched back to back on the same thin batch: is the first launch (cold code) slower?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench
from ray_tracing_amd import abi, native
scene, _ = bench.build_scene("original")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
rng = np.random.default_rng(5)
o = rng.uniform(-3, 3, (n, 3)); o[:, 1] = rng.uniform(0.2, 3, n)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
with native.Context(scene.packed) as ctx:
    do, dd = torch.tensor(o, device="cuda"), torch.tensor(d, device="cuda")
    pts = torch.empty((n, 3), dtype=torch.float64, device="cuda"); ids = torch.empty(n, dtype=torch.int64, device="cuda")
    ts = torch.cuda.Stream(); st = ts.cuda_stream
    # a frame first (other kernels run, code of k_intersect_only is cold), then the query 6 times back to back
    ctx.render(abi.render_params(500, 500, seed=1), want_rgb=False, want_ids=False)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
    ev[0].record(ts)
    for k in range(7):
        ctx.intersect_device(do.data_ptr(), dd.data_ptr(), n, pts.data_ptr(), ids.data_ptr(), stream=st)
        ev[k + 1].record(ts)
    torch.cuda.synchronize()
    print("n", n, "launch times us:", [round(ev[k].elapsed_time(ev[k + 1]) * 1e3, 1) for k in range(7)])
