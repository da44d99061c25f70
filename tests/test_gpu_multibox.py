"""Domains of several GfsBoxes on the device against the oracle WITH THE SAME BOX LAYOUT, bit for bit
(multi-box results depend on the partition: ghosts of GfsBoundaryMpi sides are lagged by one sweep,
SURVEY.md hard part 2).  Every box holds its own part of a flow with one period over the whole
lattice, so a swapped peer, a wrong rank_of or a left/right mix-up changes the result.

Several device boxes share the one GPU of the test box (one thread per box, the in-process
transport of tests/multibox.py behind the exchange / reduce hooks: RCCL refuses two ranks on one
device).  The library's own RCCL transport (csrc/transport.hip: what bench.py --gpus N uses) is
exercised on the same GPU through a one-rank communicator whose MPI sides face the box itself:
ncclSend / ncclRecv to self and a one-rank all-gather must reproduce the periodic box bit for bit.
"""
import numpy as np
import pytest

import gfship
from gfship import distributed as D
import multibox as M
from oracle import oracle as O
from flow_cases import PERIODIC, oracle_taylor_green
from test_gpu_timestep import _assert_same_state, _device_sim
from test_multibox_cpu import run_lattice_flow_threads

pytestmark = pytest.mark.gpu


counts = []      # lattice-wide coarse cycles of the boxes of the last _device_lattice_flow


def _device_lattice_flow(nboxes, level, nsteps, overlap):
    import torch
    del counts[:]
    n = 1 << level
    grid = D.BoxGrid(nboxes, 3)
    dev = torch.device("cuda", 0)

    def worker(rank, fabric):
        gd = gfship.Domain(3, level, grid.sides(rank))
        if overlap:
            gd.set_overlap(overlap)
        gs = gfship.Simulation(gd)
        hooks = D.DeviceHooks(gd, M.LocalTransport(grid, rank, fabric, dev))
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
        counts.append(gd.path_counts()[0])
        out = dict(u=gs.u[0].download()[i3], v=gs.u[1].download()[i3], w=gs.u[2].download()[i3],
                   p=gs.p.download()[i3], pmac=gs.pmac.download()[i3],
                   g=[gs.g[c].download()[i3] for c in range(3)], dt=gs.dt, t=gs.t,
                   niter=(gs.projection_params.niter, gs.approx_projection_params.niter),
                   res=gs.approx_projection_params.residual.infty)
        del hooks
        gs.destroy()
        gd.destroy()
        return out

    return M.run_boxes(nboxes, worker)


def _same(dev, ora, what):
    for rank, (d, o) in enumerate(zip(dev, ora)):
        for name in ("u", "v", "w", "p", "pmac"):
            assert np.array_equal(d[name], o[name]), (what, rank, name)
        for c in range(3):
            assert np.array_equal(d["g"][c], o["g"][c]), (what, rank, "g", c)
        assert d["dt"] == o["dt"] and d["t"] == o["t"], (what, rank)
        assert tuple(d["niter"]) == tuple(o["niter"]), (what, rank)
        assert d["res"] == o["res"], (what, rank)


@pytest.mark.parametrize("nboxes,level", [(2, 5), (4, 4), (8, 4), (8, 5)])
def test_lattice_flow_device_boxes_equal_oracle_boxes(nboxes, level):
    """overlap = 0 (plain traversal order in every sweep): 2 x 1 x 1, 2 x 2 x 1 and 2 x 2 x 2 boxes; at 32^3
    the sweeps of the three finest levels run on the pipelined tile kernel with MPI ghost streams"""
    nsteps = 2
    ora = run_lattice_flow_threads(nboxes, level, nsteps, 0)
    dev = _device_lattice_flow(nboxes, level, nsteps, 0)
    _same(dev, ora, "%d boxes" % nboxes)
    assert not np.array_equal(dev[0]["u"], dev[1]["u"])      # the boxes really differ
    # ... and the coarse ends of the V-cycles were computed for the whole lattice on every box after
    # one gather (lattice_cycle_kernel), not with one exchange per sweep
    assert len(counts) == nboxes and min(counts) >= 2 * nsteps


def test_lattice_flow_one_exchange_per_sweep_gives_the_same_bits(monkeypatch):
    """GFSHIP_NO_LATTICE_CYCLE=1: the per-level path (one halo exchange per sweep on every level)"""
    monkeypatch.setenv("GFSHIP_NO_LATTICE_CYCLE", "1")
    ora = run_lattice_flow_threads(8, 4, 2, 0)
    dev = _device_lattice_flow(8, 4, 2, 0)
    _same(dev, ora, "8 boxes, per-level exchanges")
    assert max(counts) == 0


@pytest.mark.parametrize("nboxes,level", [(2, 4), (8, 4), (8, 5)])
def test_lattice_flow_with_the_reference_default_overlap_order(nboxes, level):
    """overlap = 1, the reference's default for parallel runs (src/domain.c:682): the first
    nrelax - 1 sweeps of every relax loop visit the cells along the MPI sides first
    (src/domain.c:1093-1125), the last one is a plain traversal (src/poisson.c:1080-1086).  Device
    boxes against oracle boxes of the same lattice, bit for bit; and the iterates do differ from
    the overlap = 0 ones."""
    nsteps = 2
    ora = run_lattice_flow_threads(nboxes, level, nsteps, 1)
    dev = _device_lattice_flow(nboxes, level, nsteps, 1)
    _same(dev, ora, "%d boxes, overlap" % nboxes)
    plain = run_lattice_flow_threads(nboxes, level, nsteps, 0)
    assert not np.array_equal(plain[0]["p"], ora[0]["p"])
    assert max(counts) == 0          # the lattice-wide coarse cycle is a plain-order (overlap = 0) path


def test_rccl_transport_with_overlap_on_one_rank():
    """the split exchange of overlap = 1 (layers sent on the communicator's own stream beside the
    bulk of the sweep) through the real RCCL on a one-rank communicator, against the oracle box with
    the same MPI sides served by a mirror transport (the box is its own neighbour)"""
    from test_gpu_particles import _MirrorTransport
    level, nsteps = 5, 2
    side = [gfship.SIDE_EXTERNAL, gfship.SIDE_EXTERNAL, gfship.SIDE_PERIODIC, gfship.SIDE_PERIODIC,
            gfship.SIDE_EXTERNAL, gfship.SIDE_EXTERNAL]
    osim = O.Sim(3, level, side)
    x, y, z = osim.dom.centres()
    from flow_cases import taylor_green_3d
    for c, a in enumerate(taylor_green_3d(x, y, z)):
        osim.u[c].interior()[...] = a + (0.3 if c == 0 else 0.)
    osim.dom.set_overlap(1)

    class SelfTransport(_MirrorTransport):
        def __init__(self):
            import torch
            self.torch, self.rank = torch, 0
            self.device = torch.device("cpu")
            self._bufs = {}
            self.grid = type("G", (), {"external_sides": lambda s: [0, 1, 4, 5]})()

        def allreduce(self, vals, op):
            return np.array(vals, dtype=np.float64)

    ohooks = M.OracleHooks(O.lib(), osim.dom.ptr, 3, SelfTransport())
    gd, gs = _device_sim(osim, side)
    gd.set_overlap(1)
    gd.comm_init(gfship.comm_unique_id(), 0, 1, (1, 1, 1))
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(nsteps):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
    del ohooks
    gs.destroy()
    gd.destroy()


@pytest.mark.parametrize("axes,fast,level", [("xz", True, 5), ("xyz", True, 5), ("y", True, 5), ("xyz", False, 5),
                                             # 64^3: the Godunov sweeps along z with the states beyond the MPI
                                             # sides from the received buffers (advect3_sweep2_kernel<.., MPI>)
                                             ("xyz", True, 6), ("xz", True, 6), ("y", True, 6), ("z", True, 6)])
def test_rccl_transport_on_one_rank_reproduces_the_periodic_box(axes, fast, level, monkeypatch):
    """gfship_domain_comm_init with one rank: the sides of `axes' are GfsBoundaryMpi sides whose peer is
    the box itself, the others stay local periodic sides.  pack -> ncclSend/ncclRecv (self) -> unpack
    and the all-gather reduction, on the real RCCL, must give the periodic single box bit for bit --
    with the multi-box fast paths (coarse end of the V-cycle after one all-gather; tiled Godunov
    kernels with one message of face states per side) and without them."""
    if not fast:
        monkeypatch.setenv("GFSHIP_NO_LATTICE_CYCLE", "1")
        monkeypatch.setenv("GFSHIP_NO_FUSED_MPI", "1")
    nsteps = 2
    osim = oracle_taylor_green(level)
    osim.u[0].interior()[...] += 0.3          # something crosses the sides
    osim.u[1].interior()[...] -= 0.2
    side = [gfship.SIDE_EXTERNAL if "xyz"[d // 2] in axes else gfship.SIDE_PERIODIC for d in range(6)]
    gd, gs = _device_sim(osim, side)
    osim.set_source(1, -0.8)                  # a GfsSource on V: MAC and centred sources on both paths
    gs.set_source(1, -0.8)
    gd.comm_init(gfship.comm_unique_id(), 0, 1, (1, 1, 1))
    assert gd.comm_size() == 1
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(nsteps):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
    assert gs.approx_projection_params.residual.infty == osim.approx_projection_params.residual.infty
    for k in ("first", "second"):
        assert getattr(gs.approx_projection_params.residual, k) == pytest.approx(
            getattr(osim.approx_projection_params.residual, k), rel=1e-12)
    msgs, nbytes = gd.comm_stats()
    assert msgs > 100 and nbytes > msgs * 8
    cycles, fused = gd.path_counts()
    if fast:
        assert cycles >= 2 * nsteps and fused >= 2 * nsteps
    else:
        assert cycles == 0 and fused == 0
    gs.destroy()
    gd.destroy()


def test_spectra_of_a_lattice_of_boxes_equal_those_of_the_whole_domain():
    """GfsOutputEnergySpectra / GfsOutputSpectra on 2 x 2 x 2 boxes: every box gathers the blocks of all
    boxes (the reference redistributes slabs, modules/fft.c:467-669) and transforms the grid of the
    whole domain -- against the numpy restatement applied to the assembled field"""
    import torch
    from oracle.go_spectra import energy_spectra, output_spectra
    nboxes, level = 8, 4
    n = 1 << level
    grid = D.BoxGrid(nboxes, 3)
    dev = torch.device("cuda", 0)

    def worker(rank, fabric):
        gd = gfship.Domain(3, level, grid.sides(rank))
        hooks = D.DeviceHooks(gd, M.LocalTransport(grid, rank, fabric, dev))
        X, Y, Z = M.global_centres(grid, rank, n)
        fields, parts = [], []
        for a in M.lattice_velocity(X, Y, Z):
            b = np.zeros((n + 2,) * 3)
            b[1:-1, 1:-1, 1:-1] = a
            f = gd.variable()
            f.upload(b)
            fields.append(f)
            parts.append(np.array(a))
        out = dict(parts=parts, es=gd.energy_spectra(fields), sp=gd.output_spectra(fields[0]))
        del hooks
        gd.destroy()
        return out

    res = M.run_boxes(nboxes, worker)
    comps = [M.assemble(grid, [r["parts"][c] for r in res], n) for c in range(3)]
    k0, Ek0, Etot0 = energy_spectra(comps, n_box=n)
    F0, ks0 = output_spectra(comps[0], n_box=n)
    for r in res:
        k, Ek, Etot = r["es"]
        assert np.array_equal(k, k0)
        assert np.isclose(Etot, Etot0, rtol=1e-12)
        assert np.abs(Ek - Ek0).max() <= 1e-12 * Etot0
        F, ks = r["sp"]
        assert ks == ks0 and np.abs(F - F0).max() <= 1e-13


def test_lattice_flow_2d_device_boxes_equal_oracle_boxes():
    """2 x 2 boxes of 32^2 in 2-D: the lattice-wide coarse cycle (lattice_cycle_kernel<2>), the BC of the
    vector variables and the general Godunov kernels with MPI sides, against oracle boxes"""
    import torch
    nboxes, level, nsteps = 4, 5, 3
    n = 1 << level
    grid = D.BoxGrid(nboxes, 2)
    L = O.lib()
    dev = torch.device("cuda", 0)
    tp = 2. * np.pi

    def field(rank):
        X, Y = M.global_centres(grid, rank, n, dim=2)
        u = np.sin(tp * X) * np.cos(tp * Y) + 0.3 * np.cos(tp * (Y + 0.1)) + 0.25
        v = -np.cos(tp * X) * np.sin(tp * Y) + 0.2 * np.sin(tp * (X + 0.3))
        return u + 0. * v, v + 0. * u

    def oworker(rank, fabric):
        sim = O.Sim(2, level, grid.sides(rank))
        hooks = M.OracleHooks(L, sim.dom.ptr, 2, M.LocalTransport(grid, rank, fabric))
        for c, a in enumerate(field(rank)):
            sim.u[c].interior()[...] = a
        sim.set_time(end=10.)
        sim.start()
        for _ in range(nsteps):
            sim.step()
        out = dict(u=sim.u[0].interior().copy(), v=sim.u[1].interior().copy(), p=sim.p.interior().copy(),
                   dt=sim.dt, niter=(sim.projection_params.niter, sim.approx_projection_params.niter))
        del hooks
        return out

    def dworker(rank, fabric):
        gd = gfship.Domain(2, level, grid.sides(rank))
        gs = gfship.Simulation(gd)
        hooks = D.DeviceHooks(gd, M.LocalTransport(grid, rank, fabric, dev))
        for c, a in enumerate(field(rank)):
            b = np.zeros((n + 2,) * 2)
            b[1:-1, 1:-1] = a
            gs.u[c].upload(b)
        gs.set_time(end=10.)
        gs.start()
        for _ in range(nsteps):
            gs.step()
        gd.synchronize()
        i2 = (slice(1, -1),) * 2
        out = dict(u=gs.u[0].download()[i2], v=gs.u[1].download()[i2], p=gs.p.download()[i2], dt=gs.dt,
                   niter=(gs.projection_params.niter, gs.approx_projection_params.niter),
                   cycles=gd.path_counts()[0])
        del hooks
        gs.destroy()
        gd.destroy()
        return out

    ora = M.run_boxes(nboxes, oworker)
    dev_ = M.run_boxes(nboxes, dworker)
    for rank, (d, o) in enumerate(zip(dev_, ora)):
        for name in ("u", "v", "p"):
            assert np.array_equal(d[name], o[name]), (rank, name)
        assert d["dt"] == o["dt"] and tuple(d["niter"]) == tuple(o["niter"])
        assert d["cycles"] >= 2 * nsteps
    assert not np.array_equal(dev_[0]["u"], dev_[1]["u"])
