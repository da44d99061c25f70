"""Wall time of gfship_poisson_cycle with unit weights and with the face weights of a GfsFunction alpha
(six weight rows per cell streamed beside u / rhs / dia), periodic box: tools/weighted_cycle.py [level]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "gerris-fft-particles_amd"))
import numpy as np
import gfship
lev = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1 << lev
gd = gfship.Domain(3, lev, [gfship.SIDE_PERIODIC] * 6)
u, rhs, dia, res = (gd.variable() for _ in range(4))
rng = np.random.default_rng(0)
rhs.upload(rng.standard_normal((n + 2,) * 3))
par = gd.params()
par.depth = lev


def cycles(reps=10):
    gd.poisson_cycle(par, u, rhs, dia, res)
    gd.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gd.poisson_cycle(par, u, rhs, dia, res)
    gd.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


gd.poisson_coefficients()
print("level %d  unit weights      %.3f ms per cycle" % (lev, cycles()))
alpha = []
for c in range(3):
    a = 0.5 + rng.random((n + 2,) * 3)
    sl0, sln = [slice(None)] * 3, [slice(None)] * 3
    sl0[2 - c], sln[2 - c] = 0, -2
    a[tuple(sl0)] = a[tuple(sln)]
    v = gd.variable()
    v.upload(a)
    alpha.append(v)
gd.poisson_coefficients_alpha(alpha)
print("level %d  weights from alpha %.3f ms per cycle" % (lev, cycles()))
