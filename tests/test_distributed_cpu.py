"""N > 1 path on CPU: world_size 2 over gloo.  Each rank runs one oracle box whose x sides are
GfsBoundaryMpi sides served by gfship.distributed (the transport bench.py uses over RCCL).

Check: with every box holding one period of the same periodic field, the MPI-lagged ghost
cells equal the local periodic ghost cells of a single box at every sweep, so every rank must
reproduce the single-box run bit for bit (fields, dt, iteration counts, max residual)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from flow_cases import oracle_taylor_green
from conftest import ROOT


def test_box_lattice_topology():
    sys.path.insert(0, os.path.join(ROOT, "gerris-fft-particles_amd"))
    from gfship import distributed as D
    assert D.lattice(1) == (1, 1, 1) and D.lattice(2) == (2, 1, 1)
    assert D.lattice(4) == (2, 2, 1) and D.lattice(8) == (2, 2, 2)
    g = D.BoxGrid(8)
    assert g.sides() == [D.SIDE_EXTERNAL] * 6
    for r in range(8):
        for s in range(6):
            nb = g.neighbour(r, s)
            assert g.neighbour(nb, s ^ 1) == r       # symmetric
            assert nb != r                            # two boxes per axis: never self
    g = D.BoxGrid(2)
    assert g.sides() == [D.SIDE_EXTERNAL, D.SIDE_EXTERNAL] + [D.SIDE_PERIODIC] * 4
    assert g.neighbour(0, 0) == 1 and g.neighbour(0, 1) == 1 and g.neighbour(0, 2) == 0
    with pytest.raises(ValueError):
        D.lattice(6)


@pytest.mark.timeout(300)
def test_two_boxes_over_gloo_reproduce_single_box(tmp_path):
    level, nsteps = 4, 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(level), str(nsteps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    from oracle import oracle as O
    from particle_cases import lcg_positions
    ref = oracle_taylor_green(level)
    ref.u[0].interior()[...] += 1.       # Taylor-Green + a uniform flow along x, as in the worker
    pos, ids = lcg_positions(600)
    pos[:40, 0] = 0.5 - 1e-3 * (1 + np.arange(40))
    rpl = O.Particles(ref, pos, ids)
    ref.start()
    for _ in range(nsteps):
        rpl.event()
        ref.step()
    rpos, rid = rpl.state()
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    # particles: nothing is lost on the periodic lattice, something did cross the MPI side, both
    # boxes stay mirror images of each other, and the single periodic box agrees except for the
    # 1e-8 nudge its periodic wrap gives a crossing particle (particulatecommon.c:3204-3210)
    assert all(int(c) == 600 for d in ranks for c in d["counts"])
    assert int(ranks[0]["moved"]) > 0 and int(ranks[0]["moved"]) == int(ranks[1]["moved"])
    for d in ranks:
        o, ro = np.argsort(d["pid"]), np.argsort(rid)
        assert np.array_equal(d["pid"][o], rid[ro])
        diff = np.abs(d["ppos"][o] - rpos[ro])
        assert diff.max() < 1e-7
        assert (diff == 0.).all(axis=1).sum() > 500       # the ones that did not cross: bit equal
    o0, o1 = np.argsort(ranks[0]["pid"]), np.argsort(ranks[1]["pid"])
    assert np.array_equal(ranks[0]["ppos"][o0], ranks[1]["ppos"][o1])
    for rank in range(2):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert float(d["dt"]) == ref.dt and float(d["t"]) == ref.t
        assert int(d["niter"]) == ref.approx_projection_params.niter
        assert float(d["res"]) == ref.approx_projection_params.residual.infty
        for name, f in (("u", ref.u[0]), ("v", ref.u[1]), ("w", ref.u[2]), ("p", ref.p)):
            assert np.array_equal(d[name], f.interior()), (rank, name)
