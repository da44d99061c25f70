#!/usr/bin/env python3
"""Host-side cost of one halo exchange of gfship/distributed.py DeviceHooks without the wire:
pack kernels, bookkeeping, unpack kernels (the transport's exchange() is a no-op).
   python tools/halo_overhead.py [level]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gerris-fft-particles_amd"))
import torch
import gfship
from gfship import distributed as D
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multibox as M

level = int(sys.argv[1]) if len(sys.argv) > 1 else 8
grid = D.BoxGrid(8, 3)


class NullTransport(M.LocalTransport):
    def exchange(self, send_sides, snd, recv_sides, rcv):
        pass


dom = gfship.Domain(3, level, grid.sides(0))
tr = NullTransport(grid, 0, M.LocalFabric(1), torch.device("cuda", 0))
hooks = D.DeviceHooks(dom, tr)
v = dom.variable()
for lev in (level, 4, 0):
    n = 200
    dom.bc(v, v, lev)
    dom.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dom.bc(v, v, lev)          # -> exchange hook (all six sides) inside
    dom.synchronize()
    print("level %d: %.1f us per BC application with 6 MPI sides (no wire)"
          % (lev, (time.perf_counter() - t0) / n * 1e6))
