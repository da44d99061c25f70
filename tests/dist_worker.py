"""Worker of the world_size-2 gloo tests: one oracle box per rank with GfsBoundaryMpi sides,
driven by the same Transport / hook classes bench.py uses on GPUs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gerris-fft-particles_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from gfship import distributed as D
    from oracle import oracle as O
    from flow_cases import taylor_green_3d

    out, level, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    grid = D.BoxGrid(world, 3)
    sim = O.Sim(3, level, grid.sides(rank))
    tr = D.Transport(grid, rank)
    hooks = D.OracleHooks(O.lib(), sim.dom.ptr, 3, tr)   # keep alive
    # every box holds one period of the Taylor-Green field: the global field is continuous
    x, y, z = sim.dom.centres()
    for c, a in enumerate(taylor_green_3d(x, y, z)):
        sim.u[c].interior()[...] = a
    sim.start()
    for _ in range(nsteps):
        sim.step()
    np.savez(os.path.join(out, "rank%d.npz" % rank),
             u=sim.u[0].interior(), v=sim.u[1].interior(), w=sim.u[2].interior(),
             p=sim.p.interior(), dt=sim.dt, t=sim.t,
             niter=sim.approx_projection_params.niter,
             res=sim.approx_projection_params.residual.infty)
    del hooks
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
