"""A few exact-order relax sweeps of the 256^3 level, for counter profiling of relax_skew_kernel."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "gerris-fft-particles_amd"))
import numpy as np
import gfship
lev = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1 << lev
gd = gfship.Domain(3, lev, [gfship.SIDE_PERIODIC] * 6)
gd.poisson_coefficients()
u, rhs, dia = gd.variable(), gd.variable(), gd.variable()
rng = np.random.default_rng(0)
u.upload(rng.standard_normal((n + 2,) * 3))
rhs.upload(rng.standard_normal((n + 2,) * 3))
print("ms per sweep: %.4f" % gd.time_relax(u, rhs, dia, reps=5))
for mode, name in ((gfship.RELAX_EXACT, "fused"), (gfship.RELAX_EXACT_PER_SWEEP, "per sweep")):
    gd.set_relax_mode(mode)
    for nrelax in (2, 4):
        ms, fused = gd.time_relax_loop(u, rhs, dia, nrelax=nrelax, reps=5)
        print("relax loop nrelax %d (%s, fused=%s): %.4f ms = %.4f ms/sweep, %.0f GB/s algorithmic"
              % (nrelax, name, fused, ms, ms / nrelax, 24. * n ** 3 * nrelax / ms / 1e6))
