#!/usr/bin/env python3
"""Prints the rows of test/periodic's r0 / r1 / r2 files as computed by the quadtree oracle
(oracle/go_tree.c) next to the reference's: the long cases tests/test_oracle_tree.py leaves out
(level 7; BOX = 2 at level 6 and 7 take 1 to 10 minutes on one core).
usage: tree_ref_rows.py [boxes, e.g. 1,2] [levels, e.g. 5,6,7]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O

if len(sys.argv) > 1 and sys.argv[1] == "box":
    # test/reynolds/box: the Divergence norms of every step against div5 / div6 / div7.ref
    import numpy as np
    for level in [int(l) for l in (sys.argv[2] if len(sys.argv) > 2 else "5,6,7").split(",")]:
        ref = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "reference",
                                                    "reynolds_box_div%d.ref" % level))]
        s = O.Tree(refine=lambda x, y: level if (x > 0.25 or x < -0.25 or y > 0.25 or y < -0.25) else level + 1)
        for l in range(s.depth + 1):
            x, y = s.centres(l)
            s.values(O.Tree.U, l)[...] = - np.cos(8. * np.pi * x) * np.sin(8. * np.pi * y)
            s.values(O.Tree.V, l)[...] = np.sin(8. * np.pi * x) * np.cos(8. * np.pi * y)
        s.projection_params.tolerance = s.approx_projection_params.tolerance = 1e-6
        s.set_time(2., 0.8)
        s.start()
        k = bad = 0
        t0 = time.time()
        while True:
            f, se, inf, _ = s.divergence_norm()
            got = ("Divergence time: %g first: % 10.3e second: % 10.3e infty: % 10.3e" % (s.t, f, se, inf)).split()
            bad += got != ref[k]
            if s.t >= 2.:
                break
            s.step()
            k += 1
        print("reynolds/box LEVEL %d: %d rows, %d in the reference, %d differ (%.0f s)"
              % (level, k + 1, len(ref), bad, time.time() - t0), flush=True)
        s.destroy()
    sys.exit(0)
boxes = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2").split(",")]
levels = [int(l) for l in (sys.argv[2] if len(sys.argv) > 2 else "5,6,7").split(",")]
for box in boxes:
    ref = {}
    with open(os.path.join(ROOT, "tests", "golden", "reference", "periodic_r%d.ref" % box)) as f:
        for line in f:
            w = line.split()
            ref[int(w[0])] = w[1:3]
    for level in levels:
        t0 = time.time()
        s = O.Tree(periodic=(level, box))
        n = s.run()
        _, second, infty = s.error_norm()
        got = ["%.3e" % second, "%.3e" % infty]
        print("BOX %d LEVEL %d: oracle %s %s  reference %s %s  %s  (%d steps, %.0f s)"
              % (box, level, got[0], got[1], ref[level][0], ref[level][1],
                 "ok" if got == ref[level] else "DIFFERS", n, time.time() - t0), flush=True)
        s.destroy()
