"""Worker of tests/test_gpu_two_ranks.py: one rank of a 2 x 1 x 1 lattice of 32^3 boxes, one GPU per
rank, started under torch.distributed.run BEFORE anything touched a GPU.  torch.distributed (gloo)
carries the 128 bytes of the RCCL unique id only; halos and reductions go through the library's own
communicator over the real RCCL (csrc/transport.hip).  Writes rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gerris-fft-particles_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    out, level, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    local = int(os.environ.get("LOCAL_RANK", rank))
    import gfship
    from gfship import distributed as D
    import multibox as M
    n = 1 << level
    grid = D.BoxGrid(world, 3)
    gd = gfship.Domain(3, level, grid.sides(rank), device=local)
    ok = torch.ones(1, dtype=torch.int32)
    uid = torch.zeros(gfship.UNIQUE_ID_BYTES, dtype=torch.uint8)
    try:
        if rank == 0:
            uid = torch.frombuffer(bytearray(gfship.comm_unique_id()), dtype=torch.uint8).clone()
        else:
            gfship.comm_available()
    except Exception as e:
        sys.stderr.write("rank %d: %s\n" % (rank, e))
        ok[0] = 0
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        sys.exit(4)
    dist.broadcast(uid, 0)
    gd.comm_init(bytes(uid.numpy().tobytes()), rank, world, grid.b)
    assert gd.comm_size() == world
    gs = gfship.Simulation(gd)
    X, Y, Z = M.global_centres(grid, rank, n)
    for c, a in enumerate(M.lattice_velocity(X, Y, Z)):
        b = np.zeros((n + 2,) * 3)
        b[1:-1, 1:-1, 1:-1] = a
        gs.u[c].upload(b)
    gs.start()
    for _ in range(nsteps):
        gs.step()
    gd.synchronize()
    i3 = (slice(1, -1),) * 3
    msgs, nbytes = gd.comm_stats()
    np.savez(os.path.join(out, "rank%d.npz" % rank),
             u=gs.u[0].download()[i3], v=gs.u[1].download()[i3], w=gs.u[2].download()[i3],
             p=gs.p.download()[i3], pmac=gs.pmac.download()[i3], dt=gs.dt, t=gs.t,
             niter=np.array([gs.projection_params.niter, gs.approx_projection_params.niter]),
             res=gs.approx_projection_params.residual.infty, msgs=msgs, nbytes=nbytes)
    dist.barrier()
    gs.destroy()
    gd.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
