"""Uniformly scaled copies of the fixture scenes (x 0.001 ... x 1000): default path against the float64
reference-order walk (bit-identical float32 image, ids, ray counts) and against the oracle (ids, counters, 8-bit).
    python tools/check_scaled_vs_oracle.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_scene  # noqa: E402
import orc  # noqa: E402
from check_beam_vs_exact import scaled  # noqa: E402
from ray_tracing_amd import abi, native  # noqa: E402
from ray_tracing_amd.render import FastScene  # noqa: E402


def main():
    gold = os.path.join(ROOT, "tests", "golden", "scenes")
    texts = {n: open(os.path.join(gold, n + ".txt")).read() for n in ("pool", "original", "small", "synth400")}
    texts["synth90k"] = gen_scene.synth_mixed(300, 4321, 5, 10)
    bad_total = 0
    for name, txt0 in texts.items():
        for k in (0.001, 0.03, 1.0, 40.0, 1000.0):
            sc = FastScene(text=txt0 if k == 1.0 else scaled(txt0, k))
            ctx = native.Context(sc.packed, device=0)
            w, h = (240, 136) if name != "synth90k" else (200, 112)
            p = dict(sh_rays=3, rec_max=6, jitter=True, seed=3)
            fast = ctx.render(abi.render_params(w, h, **p))
            exact = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_EXACT_TRAVERSAL, **p))
            ref = orc.render(sc.packed, abi.render_params(w, h, **p), threads=16, want_f64=False)
            n_rgb = int((np.abs(fast["rgb"].astype(np.float64) - exact["rgb"].astype(np.float64)).max(-1) > 0).sum())
            n_ids = int((fast["ids"] != exact["ids"]).sum())
            o_ids = int((fast["ids"] != ref["ids"]).sum())
            o_rgb = int((np.abs(fast["rgb8"].astype(int) - ref["rgb8"].astype(int)).max(-1) > 1).sum())
            cnt = all(fast["stats"][c] == ref["counters"][c] for c in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits"))
            lit = float(fast["rgb"].mean())
            bad = n_rgb + n_ids + o_ids + o_rgb + (0 if cnt else 1)
            bad_total += bad
            print("%-9s x%-6g vs exact walk: rgb %d ids %d | vs oracle: ids %d rgb8>1 %d counters %s | mean %.4f" % (
                name, k, n_rgb, n_ids, o_ids, o_rgb, cnt, lit), flush=True)
            ctx.close()
    print("TOTAL:", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
