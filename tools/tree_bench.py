#!/usr/bin/env python3
"""Time steps of the refined-tree path (gfship_tree) on an octree / quadtree with BOX
extra levels inside the central cube / square: leaves, dependency levels of the finest sweep, ms per
step, Mleaf-steps/s.  Not the benchmarked path (that is bench.py); a measurement to quote.
usage: tree_bench.py [dim] [level] [box] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gerris-fft-particles_amd"))
import gfship

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3
level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
box = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
inside = lambda *a: all(-0.25 <= v <= 0.25 for v in a)
t0 = time.time()
if dim == 3:
    g = gfship.Tree(lambda x, y, z: level + box if inside(x, y, z) else level, dim=3)
else:
    g = gfship.Tree(lambda x, y: level + box if inside(x, y) else level)
t_build = time.time() - t0
nleaves = 0
for l in range(g.depth + 1):
    f = g.flags(l)
    nleaves += int(np.sum(f[(slice(1, -1),) * dim] == 1))
    c = g.centres(l)
    if dim == 3:
        x, y, z = c
        u = np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.cos(2 * np.pi * z)
        v = -np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.cos(2 * np.pi * z)
        g.upload(g.W, l, 0. * u)
    else:
        x, y = c
        u = 1. - 2. * np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y)
        v = 1. + 2. * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y)
    g.upload(g.U, l, u)
    g.upload(g.V, l, v)
g.set_time(1e30, 0.8)
g.start()
g.step()
t0 = time.time()
for _ in range(steps):
    g.step()
ms = (time.time() - t0) / steps * 1e3
nc, nl = g.sweep_levels(g.depth)
print({"dim": dim, "levels": [level, g.depth], "leaves": nleaves, "finest_sweep_cells": nc,
       "finest_sweep_dependency_levels": nl, "tree_build_s": round(t_build, 2),
       "ms_per_step": round(ms, 2), "Mleaf_steps_per_s": round(nleaves / ms / 1e3, 3),
       "niter": [g.projection_params.niter, g.approx_projection_params.niter]})
g.destroy()
