import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gerris-fft-particles_amd"))
import numpy as np, gfship
lev = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 1 << lev
gd = gfship.Domain(3, lev, [gfship.SIDE_PERIODIC]*6)
gd.poisson_coefficients()
u, rhs, dia = gd.variable(), gd.variable(), gd.variable()
rng = np.random.default_rng(0)
u.upload(rng.standard_normal((n+2,)*3)); rhs.upload(rng.standard_normal((n+2,)*3))
print("variant", os.environ.get("GFSHIP_SKEW_VARIANT", "0"), "level", lev, "ms per relax (pack+sweep+unpack): %.3f" % gd.time_relax(u, rhs, dia, reps=10))
