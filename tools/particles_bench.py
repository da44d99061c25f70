#!/usr/bin/env python3
"""Config D probe: 2M tracers on the 256^3 Taylor-Green box, particle_list_event alone.
   python tools/particles_bench.py [--level 8] [--np 2000000] [--presort]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gerris-fft-particles_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gfship
from bench import taylor_green, with_ghosts
from particle_cases import lcg_positions_fast

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=8)
ap.add_argument("--np", type=int, default=2000000)
ap.add_argument("--presort", action="store_true")
ap.add_argument("--events", type=int, default=10)
a = ap.parse_args()
n = 1 << a.level
dom = gfship.Domain(3, a.level, [gfship.SIDE_PERIODIC] * 6)
sim = gfship.Simulation(dom)
for c, f in enumerate(taylor_green(n)):
    sim.u[c].upload(with_ghosts(f))
sim.start(); sim.step()
pos, ids = lcg_positions_fast(a.np)
if a.presort:
    ijk = np.floor((pos + 0.5) * n).astype(np.int64)
    key = (ijk[:, 2] * n + ijk[:, 1]) * n + ijk[:, 0]
    o = np.argsort(key, kind="stable")
    pos, ids = pos[o], ids[o]
pl = gfship.ParticleList(sim, pos, ids)
pl.event(); dom.synchronize()
t0 = time.perf_counter()
for _ in range(a.events):
    pl.event()
dom.synchronize()
dt = (time.perf_counter() - t0) / a.events
print("np %d level %d presort %s: %.3f ms/event, %.1f Mparticle-steps/s, alive %d"
      % (a.np, a.level, a.presort, dt * 1e3, a.np / dt / 1e6, pl.count()))
