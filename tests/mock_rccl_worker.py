"""Worker of tests/test_gpu_mock_rccl.py: runs in a process of its own with GFSHIP_RCCL_LIBRARY naming
tests/mock_rccl/librccl_mock.so (libgfship opens its RCCL once per process).  N ranks of the library's
OWN transport (csrc/transport.hip: comm_exchange kind 0 and 1 + e, comm_exchange_begin / _end,
comm_exchange_raw, comm_reduce, comm_allgather, comm_migrate) as N threads on one GPU, every box
holding its part of a field with one period over the lattice, against oracle boxes of the same
lattice, bit for bit.  Prints one JSON line.

  python tests/mock_rccl_worker.py flow NBOXES LEVEL NSTEPS OVERLAP [nofast]
  python tests/mock_rccl_worker.py particles
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "librccl_mock.so")
os.environ["GFSHIP_RCCL_LIBRARY"] = MOCK
for p in (ROOT, os.path.join(ROOT, "gerris-fft-particles_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

import gfship  # noqa: E402
from gfship import distributed as D  # noqa: E402
import multibox as M  # noqa: E402


def mock_counters():
    L = C.CDLL(MOCK)      # the copy libgfship opened (same path: same handle)
    for f in ("mock_rccl_mismatches", "mock_rccl_messages", "mock_rccl_timeouts"):
        getattr(L, f).restype = C.c_ulonglong
    return dict(mismatches=int(L.mock_rccl_mismatches()), messages=int(L.mock_rccl_messages()),
                timeouts=int(L.mock_rccl_timeouts()))


def flow(nboxes, level, nsteps, overlap, fast):
    from test_multibox_cpu import run_lattice_flow_threads
    if not fast:
        os.environ["GFSHIP_NO_LATTICE_CYCLE"] = "1"
        os.environ["GFSHIP_NO_FUSED_MPI"] = "1"
    n = 1 << level
    grid = D.BoxGrid(nboxes, 3)
    uid = gfship.comm_unique_id()

    def worker(rank, fabric):
        gd = gfship.Domain(3, level, grid.sides(rank))
        if overlap:
            gd.set_overlap(overlap)
        gd.comm_init(uid, rank, nboxes, grid.b)
        assert gd.comm_size() == nboxes
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
        out = dict(u=gs.u[0].download()[i3], v=gs.u[1].download()[i3], w=gs.u[2].download()[i3],
                   p=gs.p.download()[i3], pmac=gs.pmac.download()[i3],
                   g=[gs.g[c].download()[i3] for c in range(3)], dt=gs.dt, t=gs.t,
                   niter=(gs.projection_params.niter, gs.approx_projection_params.niter),
                   res=gs.approx_projection_params.residual.infty,
                   counts=gd.path_counts(), stats=gd.comm_stats())
        fabric.barrier.wait()       # nobody destroys buffers a neighbour may still read
        gs.destroy()
        gd.destroy()
        return out

    dev = M.run_boxes(nboxes, worker)
    ora = run_lattice_flow_threads(nboxes, level, nsteps, overlap)
    bad = []
    for rank, (d, o) in enumerate(zip(dev, ora)):
        for name in ("u", "v", "w", "p", "pmac"):
            if not np.array_equal(d[name], o[name]):
                bad.append((rank, name))
        for c in range(3):
            if not np.array_equal(d["g"][c], o["g"][c]):
                bad.append((rank, "g%d" % c))
        if d["dt"] != o["dt"] or d["t"] != o["t"] or tuple(d["niter"]) != tuple(o["niter"]) or \
                d["res"] != o["res"]:
            bad.append((rank, "scalars"))
    return dict(bad=bad, differ=bool(not np.array_equal(dev[0]["u"], dev[1]["u"])),
                lattice_cycles=[int(d["counts"][0]) for d in dev],
                fused_mpi=[int(d["counts"][1]) for d in dev],
                sent=[int(d["stats"][0]) for d in dev], **mock_counters())


def particles():
    """tracers crossing the sides of 2 x 2 x 2 boxes: comm_migrate with real peers (counts, then the
    records; sides without leavers included) against oracle boxes of the same lattice that hand their
    packets over through the in-process transport (send_particles / rcv_particles,
    modules/particulatecommon.c:3218-3312)"""
    from oracle import oracle as O
    from particle_cases import lcg_positions
    nboxes, level, nev = 8, 4, 6
    n = 1 << level
    grid = D.BoxGrid(nboxes, 3)
    uid = gfship.comm_unique_id()
    L = O.lib()

    def seeds(rank):
        pos, ids = lcg_positions(300, seed=17 + rank)
        return pos, (ids + 1000 * rank).astype(np.uint32)

    def velocity(rank):
        X, Y, Z = M.global_centres(grid, rank, n)
        u, v, w = M.lattice_velocity(X, Y, Z)
        return [2. * u + 0.8, 2. * v - 0.6, 2. * w + 0.7]       # through-flow across every side

    def oworker(rank, fabric):
        sim = O.Sim(3, level, grid.sides(rank))
        tr = M.LocalTransport(grid, rank, fabric)
        hooks = M.OracleHooks(L, sim.dom.ptr, 3, tr)
        for c, a in enumerate(velocity(rank)):
            sim.u[c].interior()[...] = a
        opl = O.Particles(sim, *seeds(rank))
        sim.start()
        moved = 0
        for _ in range(nev):
            opl.event()
            out = {d: opl.outbox(d) for d in grid.external_sides()}
            moved += sum(len(a) for a in out.values())
            opl.clear_outbox()
            for d, a in sorted(tr.exchange_records(out).items()):
                opl.append(a)
            sim.step()
        st = opl.state()
        del hooks
        return st + (moved,)

    def dworker(rank, fabric):
        gd = gfship.Domain(3, level, grid.sides(rank))
        gd.comm_init(uid, rank, nboxes, grid.b)
        gs = gfship.Simulation(gd)
        for c, a in enumerate(velocity(rank)):
            b = np.zeros((n + 2,) * 3)
            b[1:-1, 1:-1, 1:-1] = a
            gs.u[c].upload(b)
        pl = gfship.ParticleList(gs, *seeds(rank))
        pl.set_sort_interval(2)
        gs.start()
        for _ in range(nev):
            pl.event()
            gs.step()
        gd.synchronize()
        p, i = pl.download()
        fabric.barrier.wait()
        pl.destroy()
        gs.destroy()
        gd.destroy()
        return p, i

    dev = M.run_boxes(nboxes, dworker)
    ora = M.run_boxes(nboxes, oworker)
    bad, moved, foreign = [], 0, 0
    for rank, ((dp, di), (op, oi, mv)) in enumerate(zip(dev, ora)):
        a, b = np.argsort(di, kind="stable"), np.argsort(oi, kind="stable")
        if not (len(di) == len(oi) and np.array_equal(di[a], oi[b]) and np.array_equal(dp[a], op[b])):
            bad.append(rank)
        moved += int(mv)
        foreign += int(np.sum(np.asarray(oi) // 1000 != rank))
    return dict(bad=bad, moved=moved, foreign=foreign, **mock_counters())


if __name__ == "__main__":
    if sys.argv[1] == "flow":
        out = flow(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
                   len(sys.argv) < 7 or sys.argv[6] != "nofast")
    else:
        out = particles()
    print("MOCKRCCL " + json.dumps(out))
