"""Frames of many workloads, the default path (beam verdicts, float32 filtered predicates, pipelined levels) against
RT_FLAG_NO_BEAM (every shadow ray decided by the float64 walk): the float32 images must be identical.  Includes
uniformly scaled copies of scenes (coordinates x 0.01 ... x 100): every error bound in the filters has to scale.

    python tools/check_beam_vs_exact.py [--quick]"""
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


def scaled(text, k):
    """Every length of a scene file times k (positions, radii, box sizes, plane offsets, light radii, camera)."""
    out = []
    for line in text.splitlines():
        t = line.split()
        if not t or t[0].startswith("#"):
            out.append(line)
            continue
        v = [float(x) for x in t[1:]]
        if t[0] == "cam":
            idx = [0, 1, 2, 3, 4, 5, 9, 10]
        elif t[0] == "sph":
            idx = [0, 1, 2, 3]
        elif t[0] == "box":
            idx = [0, 1, 2, 3]
        elif t[0] == "pln":
            idx = [3]
        elif t[0] == "lgt":
            idx = [0, 1, 2, 8]
        else:
            idx = []
        for i in idx:
            v[i] *= k
        if t[0] in ("sph", "box", "pln"):
            v[-1] = int(v[-1])
        out.append(t[0] + " " + " ".join(repr(x) for x in v))
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    gold = os.path.join(ROOT, "tests", "golden", "scenes")
    texts = {
        "pool": open(os.path.join(gold, "pool.txt")).read(),
        "original": open(os.path.join(gold, "original.txt")).read(),
        "small": open(os.path.join(gold, "small.txt")).read(),
        "synth10k": gen_scene.synth_spheres(100, 1234, 5, 5),
        "synth90k": gen_scene.synth_mixed(300, 4321, 5, 10),
    }
    if not a.quick:
        texts["synth1M"] = gen_scene.synth_mixed(1000, 4321, 5, 10)
    cases = []
    for name in texts:
        cases.append((name, 1.0, 1920, 1080, None))
    for name in ("pool", "original", "small", "synth10k", "synth90k"):
        for k in (0.01, 0.1, 10.0, 100.0, 1000.0):
            cases.append((name, k, 640, 360, 3))
    cases.append(("synth10k", 1.0, 3840, 2160, None))
    total_bad = 0
    for name, k, w, h, sh in cases:
        txt = texts[name] if k == 1.0 else scaled(texts[name], k)
        sc = FastScene(text=txt)
        ctx = native.Context(sc.packed, device=0)
        for seed in (11, 12):
            kw = dict(jitter=True, seed=seed, sh_rays=sh)
            fast = ctx.render(abi.render_params(w, h, **kw))
            exact = ctx.render(abi.render_params(w, h, flags=abi.RT_FLAG_NO_BEAM, **kw))
            d = np.abs(fast["rgb"].astype(np.float64) - exact["rgb"].astype(np.float64)).max(-1)
            bad = np.nonzero(d > 0)[0]
            ids_ok = np.array_equal(fast["ids"], exact["ids"])
            total_bad += len(bad) + (0 if ids_ok else 1)
            print("%-9s x%-6g %4dx%-4d seed %d  pixels differ: %d %s ids %s   %.2f / %.2f ms" % (
                name, k, w, h, seed, len(bad), bad[:6], ids_ok, fast["stats"]["ms_total"], exact["stats"]["ms_total"]), flush=True)
        ctx.close()
    print("TOTAL differing pixels:", total_bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
