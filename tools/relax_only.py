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
