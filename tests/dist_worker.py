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
    import multibox as M
    from oracle import oracle as O
    from flow_cases import taylor_green_3d

    out, level, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    field = sys.argv[4] if len(sys.argv) > 4 else "period"
    overlap = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    grid = D.BoxGrid(world, 3)
    sim = O.Sim(3, level, grid.sides(rank))
    sim.dom.set_overlap(overlap)
    tr = D.Transport(grid, rank)
    hooks = M.OracleHooks(O.lib(), sim.dom.ptr, 3, tr)   # keep alive
    if field == "lattice":
        # every box holds ITS part of a flow with one period over the whole lattice
        X, Y, Z = M.global_centres(grid, rank, 1 << level)
        for c, a in enumerate(M.lattice_velocity(X, Y, Z)):
            sim.u[c].interior()[...] = a
        sim.start()
        for _ in range(nsteps):
            sim.step()
        np.savez(os.path.join(out, "rank%d.npz" % rank),
                 u=sim.u[0].interior(), v=sim.u[1].interior(), w=sim.u[2].interior(),
                 p=sim.p.interior(), dt=sim.dt, t=sim.t)
        del hooks
        dist.destroy_process_group()
        return
    # every box holds one period of the Taylor-Green field: the global field is continuous
    x, y, z = sim.dom.centres()
    for c, a in enumerate(taylor_green_3d(x, y, z)):
        sim.u[c].interior()[...] = a + (1. if c == 0 else 0.)    # + a uniform flow through the MPI sides
    # tracers: every box starts with the same set; what leaves through an MPI side goes to the
    # neighbour box (the same transport moves the packets)
    from particle_cases import lcg_positions
    pos, ids = lcg_positions(600)
    pos[:40, 0] = 0.5 - 1e-3 * (1 + np.arange(40))       # some right at the +x MPI side
    pl = O.Particles(sim, pos, ids)
    counts, moved = [], 0
    sim.start()
    for _ in range(nsteps):
        pl.event()                                        # events first (simulation.c:483)
        packets = {d: pl.outbox(d) for d in grid.external_sides()}
        moved += sum(len(a) for a in packets.values())
        pl.clear_outbox()
        inc = tr.exchange_records(packets)
        for d in sorted(inc):
            pl.append(inc[d])
        counts.append(pl.count())
        sim.step()
    ppos, pid = pl.state()
    np.savez(os.path.join(out, "rank%d.npz" % rank),
             ppos=ppos, pid=pid, counts=np.array(counts), moved=moved,
             u=sim.u[0].interior(), v=sim.u[1].interior(), w=sim.u[2].interior(),
             p=sim.p.interior(), dt=sim.dt, t=sim.t,
             niter=sim.approx_projection_params.niter,
             res=sim.approx_projection_params.residual.infty)
    del hooks
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
