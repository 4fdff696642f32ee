#!/usr/bin/env python3
"""Synthetic scene generators (inputs 3-5).

Scenes are emitted in the reference's plain-text scene format (README of the
reference, `Parser.parse_scene_file`, Parser.py:25-71) so that the reference
parser, the host mirror in `ray-tracing_amd/host/Parser.py` and the oracle all
read exactly the same bytes.  `lgt` lines always come last and
no point of space is covered by more than two primitives (Q9).

    python tools/gen_scene.py synth --grid 100 --seed 1234 -o /tmp/synth10k.txt
    python tools/gen_scene.py mixed --grid 1000 --seed 4321 -o /tmp/synth1M.txt
"""
from __future__ import annotations

import argparse
import io

import numpy as np


def _fmt(vals):
    return " ".join(repr(float(v)) for v in vals)


def synth_spheres(grid: int = 100, seed: int = 1234, sh_rays: int = 5, rec_max: int = 5,
                  walls: bool = False) -> str:
    """`grid`x`grid` jittered lattice of spheres over a ground plane (config 3/4).

    grid=100 gives the 10 k-sphere scene BASELINE.json's metric is quoted on.
    """
    rng = np.random.default_rng(seed)
    out = io.StringIO()
    half = grid / 2.0
    # camera scaled with the lattice so every size frames the whole field
    k = grid / 100.0
    out.write("# synthetic sphere lattice: grid=%d seed=%d\n" % (grid, seed))
    out.write("cam %s %s %s 1.4 1\n" % (_fmt([0.0, 30.0 * k, -80.0 * k]), _fmt([0.0, 0.0, 0.0]),
                                         _fmt([0.0, 1.0, 0.0])))
    out.write("set 0.5 0.7 1.0 %d %d\n" % (sh_rays, rec_max))
    n_mtl = 16
    for m in range(n_mtl):
        kd = rng.uniform(0.1, 0.95, 3)
        ks = rng.uniform(0.2, 1.0, 3)
        kr = rng.uniform(0.1, 0.4, 3) if m % 4 == 0 else np.zeros(3)
        phong = int(rng.integers(5, 60))
        out.write("mtl %s %s %s %d 0\n" % (_fmt(kd), _fmt(ks), _fmt(kr), phong))
    out.write("mtl 0.45 0.45 0.4 0 0 0 0 0 0 1 0\n")          # 17: matte ground
    out.write("pln 0 1 0 -1 17\n")
    if walls:
        out.write("pln 0 0 -1 %s 17\n" % repr(-(half + 5.0)))
    for ix in range(grid):
        for iz in range(grid):
            cx = ix - half + 0.5 + rng.uniform(-0.15, 0.15)
            cz = iz - half + 0.5 + rng.uniform(-0.15, 0.15)
            r = rng.uniform(0.15, 0.33)
            cy = -1.0 + r + rng.uniform(0.0, 2.0)
            mat = 1 + int(rng.integers(0, n_mtl))
            out.write("sph %s %s %d\n" % (_fmt([cx, cy, cz]), repr(float(r)), mat))
    for lx, lz in ((0.0, 0.0), (-0.3, -0.3), (-0.3, 0.3), (0.3, -0.3), (0.3, 0.3)):
        out.write("lgt %s 0.5 0.5 0.45 1 0.9 2\n" % _fmt([lx * grid, 40.0, lz * grid]))
    return out.getvalue()


def synth_mixed(grid: int = 1000, seed: int = 4321, sh_rays: int = 5, rec_max: int = 10) -> str:
    """`grid`x`grid` lattice alternating spheres and boxes, 25 % reflective and
    15 % transparent materials (config 5; grid=1000 gives 1 M primitives)."""
    rng = np.random.default_rng(seed)
    out = io.StringIO()
    half = grid / 2.0
    k = grid / 100.0
    out.write("# synthetic mixed sph/box lattice: grid=%d seed=%d\n" % (grid, seed))
    out.write("cam %s %s %s 1.4 1\n" % (_fmt([0.0, 30.0 * k, -80.0 * k]), _fmt([0.0, 0.0, 0.0]),
                                         _fmt([0.0, 1.0, 0.0])))
    out.write("set 0.5 0.7 1.0 %d %d\n" % (sh_rays, rec_max))
    n_mtl = 20
    for m in range(n_mtl):
        kd = rng.uniform(0.1, 0.95, 3)
        ks = rng.uniform(0.2, 1.0, 3)
        kr = rng.uniform(0.1, 0.4, 3) if m % 4 == 0 else np.zeros(3)          # 25 % reflective
        tr = float(rng.uniform(0.3, 0.9)) if m % 20 in (1, 6, 11) else 0.0    # 15 % transparent
        phong = int(rng.integers(5, 60))
        out.write("mtl %s %s %s %d %s\n" % (_fmt(kd), _fmt(ks), _fmt(kr), phong, repr(tr)))
    out.write("mtl 0.45 0.45 0.4 0 0 0 0 0 0 1 0\n")
    out.write("pln 0 1 0 -1 %d\n" % (n_mtl + 1))
    kinds = rng.integers(0, 2, size=(grid, grid))
    jit = rng.uniform(-0.12, 0.12, size=(grid, grid, 2))
    size = rng.uniform(0.0, 1.0, size=(grid, grid))
    lift = rng.uniform(0.0, 2.0, size=(grid, grid))
    mats = 1 + rng.integers(0, n_mtl, size=(grid, grid))
    for ix in range(grid):
        for iz in range(grid):
            cx = ix - half + 0.5 + jit[ix, iz, 0]
            cz = iz - half + 0.5 + jit[ix, iz, 1]
            if kinds[ix, iz]:
                r = 0.15 + 0.18 * size[ix, iz]
                out.write("sph %s %s %d\n" % (_fmt([cx, -1.0 + r + lift[ix, iz], cz]), repr(float(r)),
                                               mats[ix, iz]))
            else:
                sc = 0.3 + 0.3 * size[ix, iz]
                out.write("box %s %s %d\n" % (_fmt([cx, -1.0 + sc / 2 + lift[ix, iz], cz]), repr(float(sc)),
                                               mats[ix, iz]))
    for lx, lz in ((0.0, 0.0), (-0.3, -0.3), (-0.3, 0.3), (0.3, -0.3), (0.3, 0.3)):
        out.write("lgt %s 0.5 0.5 0.45 1 0.9 2\n" % _fmt([lx * grid, 40.0, lz * grid]))
    return out.getvalue()


def small_mixed(seed: int = 7, n: int = 200) -> str:
    """~n spheres/boxes in a box-shaped volume with reflective *and* transparent
    materials, two planes, a light with shadow_intensity 0 and a light placed so
    that a light->hit direction with z == 0 occurs (Q7).  Small enough that the
    reference renders it in seconds; used for golden fixtures."""
    rng = np.random.default_rng(seed)
    out = io.StringIO()
    out.write("# small mixed scene: n=%d seed=%d\n" % (n, seed))
    out.write("cam 0.3 2.5 -9 0 0.4 0 0 1 0 1.2 1\n")
    out.write("set 0.1 0.2 0.3 2 4\n")
    out.write("mtl 0.8 0.2 0.2 0.9 0.9 0.9 0.3 0.3 0.3 20.7 0\n")     # non-integer shininess (Q10)
    out.write("mtl 0.2 0.8 0.2 0.5 0.5 0.5 0 0 0 7 0.6\n")            # transparent
    out.write("mtl 0.2 0.3 0.9 1 1 1 0.2 0.2 0.5 41 0.4\n")           # transparent + reflective
    out.write("mtl 0.7 0.7 0.2 0.3 0.3 0.3 0 0 0 3 0\n")
    out.write("mtl 0.6 0.6 0.6 0 0 0 0.15 0.15 0.15 1 0\n")           # ground
    out.write("mtl 0.3 0.5 0.7 0.2 0.2 0.2 0 0 0 0 0\n")              # wall, shininess 0 (0**0)
    out.write("pln 0 1 0 -1 5\n")
    out.write("pln 0 0 -1 -6 6\n")
    side = int(np.ceil(n ** (1 / 3)))
    cnt = 0
    for ix in range(side):
        for iy in range(side):
            for iz in range(side):
                if cnt >= n:
                    break
                c = np.array([ix - side / 2 + 0.5, iy * 0.9 - 0.4, iz - side / 2 + 0.5]) * 0.8
                c += rng.uniform(-0.08, 0.08, 3)
                mat = 1 + int(rng.integers(0, 4))
                if rng.uniform() < 0.5:
                    out.write("sph %s %s %d\n" % (_fmt(c), repr(float(rng.uniform(0.12, 0.3))), mat))
                else:
                    out.write("box %s %s %d\n" % (_fmt(c), repr(float(rng.uniform(0.2, 0.5))), mat))
                cnt += 1
    out.write("lgt 0 6 -4 0.9 0.9 0.8 1 0.8 1.5\n")
    out.write("lgt -5 4 -2 0.4 0.4 0.6 0.5 0 1\n")                    # shadow_intensity 0: no shadow rays
    out.write("lgt 4 -1 -5 0.3 0.3 0.3 1 0.6 0.5\n")                  # y == ground plane height
    return out.getvalue()


def on_cut_symmetric() -> str:
    """A scene that is mirror-symmetric about x = 0 with the camera and one light ON that plane: the root's cut lies at
    x = 0 exactly, the centre column of an odd-width frame has direction.x == 0 exactly, and so have the shadow rays from
    the on-axis light to points of that column.  The reference then evaluates 0 * inf = NaN in its slab test
    (Cube.py:120-136) and culls both children of the root for those rays - an input where "the tree walk equals brute
    force" does not hold.  Deterministic (no random numbers)."""
    out = io.StringIO()
    out.write("# mirror-symmetric scene: camera, light 1 and the root cut on the plane x = 0\n")
    out.write("cam 0 2 -8 0 0 4 0 1 0 1.5 1\n")
    out.write("set 0.25 0.35 0.5 3 3\n")
    out.write("mtl 0.8 0.3 0.2 0.8 0.8 0.8 0.25 0.25 0.25 25 0\n")
    out.write("mtl 0.2 0.7 0.3 0.5 0.5 0.5 0 0 0 9 0\n")
    out.write("mtl 0.3 0.4 0.9 0.9 0.9 0.9 0 0 0 40 0.5\n")
    out.write("mtl 0.6 0.6 0.55 0.1 0.1 0.1 0.1 0.1 0.1 2 0\n")
    out.write("pln 0 1 0 -1 4\n")
    k = 0
    for iz in range(6):
        for ix, x in enumerate((0.75, 1.75, 2.75)):
            y = -1.0 + 0.375 + 0.25 * ((ix + iz) % 3)
            r = 0.25 + 0.125 * ((ix + 2 * iz) % 2)
            for sx in (-1.0, 1.0):
                if (ix + iz) % 4 == 3:
                    out.write("box %r %r %r %r %d\n" % (sx * x, y, float(iz), 2 * r, 1 + k % 3))
                else:
                    out.write("sph %r %r %r %r %d\n" % (sx * x, y, float(iz), r, 1 + k % 3))
            k += 1
    out.write("sph 0 -0.25 2.5 0.5 1\n")          # straddles the plane x = 0: geometrically in front of the centre column
    out.write("box 0 -0.5 5 0.75 2\n")
    out.write("lgt 0 7 1 0.9 0.9 0.85 1 0.8 1.5\n")
    out.write("lgt -4 5 -3 0.4 0.4 0.5 1 0.6 1\n")
    return out.getvalue()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("kind", choices=["synth", "mixed", "small"])
    ap.add_argument("--grid", type=int, default=100)
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--sh-rays", type=int, default=5)
    ap.add_argument("--rec-max", type=int, default=None)
    ap.add_argument("-o", "--output", required=True)
    a = ap.parse_args()
    if a.kind == "synth":
        txt = synth_spheres(a.grid, 1234 if a.seed is None else a.seed, a.sh_rays, 5 if a.rec_max is None else a.rec_max)
    elif a.kind == "mixed":
        txt = synth_mixed(a.grid, 4321 if a.seed is None else a.seed, a.sh_rays, 10 if a.rec_max is None else a.rec_max)
    else:
        txt = small_mixed(7 if a.seed is None else a.seed, a.n)
    with open(a.output, "w") as f:
        f.write(txt)


if __name__ == "__main__":
    main()
