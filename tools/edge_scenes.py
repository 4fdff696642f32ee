"""Degenerate scene files through the whole path, against the oracle: no finite primitive, no light, no surface at all,
only shadowless lights, a single primitive, a camera inside a sphere / box.    python tools/edge_scenes.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
from ray_tracing_amd import abi, native  # noqa: E402
from ray_tracing_amd.render import FastScene  # noqa: E402

HEAD = "cam 0 2 -6 0 0 0 0 1 0 1.2 1\nset 0.3 0.5 0.8 3 4\nmtl 0.8 0.2 0.2 1 1 1 0.3 0.3 0.3 20 0\nmtl 0.2 0.8 0.2 0.5 0.5 0.5 0 0 0 5 0.6\n"
CASES = {
    "plane and light only": HEAD + "pln 0 1 0 -1 1\nlgt 0 5 0 1 1 1 1 0.8 1\n",
    "no light": HEAD + "pln 0 1 0 -1 1\nsph 0 0 0 1 2\n",
    "nothing but the background": HEAD,
    "lights only": HEAD + "lgt 0 5 0 1 1 1 1 0.8 1\nlgt 3 5 0 1 1 1 1 0 1\n",
    "shadowless lights": HEAD + "pln 0 1 0 -1 1\nsph 0 0 0 1 1\nbox 2 0 0 1 2\nlgt 0 5 0 1 1 1 1 0 1\nlgt 3 5 -2 1 1 1 0.5 0 2\n",
    "one sphere": HEAD + "sph 0 0 0 1 1\nlgt 0 5 -3 1 1 1 1 1 0.5\n",
    "one box": HEAD + "box 0 0 0 1.5 2\nlgt 0 5 -3 1 1 1 1 1 0.5\n",
    "camera inside a sphere": HEAD.replace("cam 0 2 -6", "cam 0 0.1 -0.2") + "sph 0 0 0 3 2\nsph 0 0 1 0.3 1\nlgt 0 1 0 1 1 1 1 0.7 0.3\n",
    "camera inside a box": HEAD.replace("cam 0 2 -6", "cam 0 0.1 -0.2") + "box 0 0 0 6 2\nsph 0 0 1 0.3 1\nlgt 0 1 0 1 1 1 1 0.7 0.3\n",
    "light on a surface": HEAD + "pln 0 1 0 -1 1\nsph 0 0 0 1 1\nlgt 0 -1 2 1 1 1 1 0.9 1\nlgt 0 1 0 1 1 1 1 0.9 0.5\n",
    "zero radius light": HEAD + "pln 0 1 0 -1 1\nsph 0 0 0 1 1\nlgt 2 5 -2 1 1 1 1 0.9 0\n",
    "material index past the table": HEAD + "sph 0 0 0 1 7\nlgt 0 5 -3 1 1 1 1 1 0.5\n",
}


def main():
    total = 0
    for name, txt in CASES.items():
        try:
            sc = FastScene(text=txt)
            ctx = native.Context(sc.packed, device=0)
        except Exception as e:
            print("%-32s refused: %s" % (name, e), flush=True)
            continue
        p = abi.render_params(96, 64, jitter=True, seed=4)
        out = ctx.render(p)
        ex = ctx.render(abi.render_params(96, 64, jitter=True, seed=4, flags=abi.RT_FLAG_EXACT_TRAVERSAL))
        ref = orc.render(sc.packed, p, threads=8, want_f64=False)
        bad = int((out["ids"] != ref["ids"]).sum()) + int((np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max(-1) > 1).sum())
        bad += int((np.abs(out["rgb"].astype(np.float64) - ex["rgb"].astype(np.float64)).max(-1) > 0).sum())
        bad += 0 if all(out["stats"][c] == ref["counters"][c] for c in ("primary_rays", "shadow_rays", "reflect_rays", "transmit_rays", "hits")) else 1
        total += bad
        print("%-32s differences %d   rays %d  mean %.4f%s" % (name, bad, out["stats"]["rays"], float(np.nanmean(out["rgb"])), "  <<<<" if bad else ""), flush=True)
        ctx.close()
    print("TOTAL:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
