"""Large lattices (tools/gen_scene.synth_mixed, 90 k - 360 k primitives) with the lights moved to random places, among
them a few degrees above the horizon of far points: default path against the float64 shadow walk, bit for bit.
    [RT_AMD_BEAM_EARLY=0|1] python tools/fuzz_lattice.py [--n 8]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scene  # noqa: E402
from ray_tracing_amd import abi, native  # noqa: E402
from ray_tracing_amd.render import FastScene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    a = ap.parse_args()
    total = 0
    for i in range(a.n):
        rng = np.random.default_rng(100 + i)
        grid = int(rng.choice([300, 400, 600]))
        lines = [l for l in gen_scene.synth_mixed(grid, 1000 + i, 5, 10).splitlines() if not l.startswith("lgt")]
        for _ in range(3):
            r = grid * rng.uniform(0.0, 1.5)
            ang = rng.uniform(0, 2 * np.pi)
            y = float(rng.choice([40.0, 5.0, 1.5, 200.0]))
            lines.append("lgt %r %r %r 0.5 0.5 0.45 1 0.9 %r" % (float(r * np.cos(ang)), y, float(r * np.sin(ang)), float(10 ** rng.uniform(-1, 1))))
        sc = FastScene(text="\n".join(lines) + "\n")
        ctx = native.Context(sc.packed, device=0)
        for seed in (1, 2):
            kw = dict(jitter=True, seed=seed, sh_rays=int(rng.choice([3, 4, 5])))
            fast = ctx.render(abi.render_params(640, 360, **kw))
            exact = ctx.render(abi.render_params(640, 360, flags=abi.RT_FLAG_NO_BEAM, **kw))
            bad = int((np.abs(fast["rgb"].astype(np.float64) - exact["rgb"].astype(np.float64)).max(-1) > 0).sum()) + int((fast["ids"] != exact["ids"]).sum())
            total += bad
            print("lattice %d grid %d seed %d s %d: pixels differ %d   %.2f / %.2f ms%s" % (i, grid, seed, kw["sh_rays"], bad, fast["stats"]["ms_total"],
                                                                                         exact["stats"]["ms_total"], "  <<<<<<" if bad else ""), flush=True)
        ctx.close()
    print("TOTAL:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
