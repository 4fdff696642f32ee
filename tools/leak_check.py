"""Create / render / destroy contexts and scenes in a loop and watch the device's free memory (torch.cuda.mem_get_info)
and the process RSS: neither may drift.    python tools/leak_check.py"""
import os
import sys

import numpy as np
import psutil
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scene  # noqa: E402
from ray_tracing_amd import abi, native  # noqa: E402
from ray_tracing_amd.render import FastScene  # noqa: E402

text = gen_scene.synth_mixed(100, 1, 3, 6)
proc = psutil.Process()


def one(i):
    sc = FastScene(text=text)
    ctx = native.Context(sc.packed, device=0)
    out = ctx.render(abi.render_params(160 + i % 3 * 64, 90, jitter=True, seed=i))
    for k in range(3):                      # frames in flight: the second copy of the level-0 state is made and freed too
        ctx.render_device(abi.render_params(160, 90, jitter=True, seed=k))
    ctx.sync_stats()
    assert native.lib().rt_debug_early_frames(ctx._h) >= 1
    ctx.render_mean(abi.render_params(64, 48, jitter=True, seed=i), 3)
    o = np.random.default_rng(i).normal(size=(1000, 3)); d = np.random.default_rng(i + 1).normal(size=(1000, 3))
    ctx.intersect(o, d); ctx.visibility(o, d, o); ctx.trace(abi.render_params(1, 1, sh_rays=3, rec_max=2), o, d)
    c = np.random.default_rng(i).uniform(0, 10, (500, 3))
    native.bsp_build(c, c + 0.2)
    ctx.close()
    sc.packed.close()
    return out["stats"]["rays"]


for i in range(5):
    one(i)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]; rss0 = proc.memory_info().rss
for i in range(60):
    one(i)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]; rss1 = proc.memory_info().rss
print("device free: %.1f -> %.1f MB (drift %.2f MB) | RSS %.1f -> %.1f MB (drift %.2f MB) over 60 rounds" % (
    free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6, rss0 / 1e6, rss1 / 1e6, (rss1 - rss0) / 1e6))
