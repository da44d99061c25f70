"""GPU parity of the Lagrangian tracer path against the oracle: positions
bit-identical, hence cell indices identical, ids and survivor lists identical."""
import numpy as np
import pytest

import gfship
from flow_cases import PERIODIC, oracle_taylor_green, oracle_reynolds
from oracle import oracle as O
from particle_cases import lcg_positions
from test_gpu_timestep import _device_sim

pytestmark = pytest.mark.gpu


def _run(osim, side, pos, ids, nsteps, with_flow_steps=True, end=None):
    gd, gs = _device_sim(osim, side)
    if end is not None:
        gs.set_time(end=end)
    osim.start()
    gs.start()
    opl = O.Particles(osim, pos, ids)
    gpl = gfship.ParticleList(gs, pos, ids)
    for k in range(nsteps):
        # simulation_run: events (particles) first, then the flow step (simulation.c:483)
        opl.event()
        gpl.event()
        op, oi = opl.state()
        gp, gi = gpl.download()
        assert np.array_equal(oi, gi), k
        assert np.array_equal(op, gp), (k, np.abs(op - gp).max())
        assert gpl.count() == len(oi)
        if with_flow_steps:
            osim.step()
            gs.step()
    return opl, gpl


def test_tracers_in_taylor_green_3d_bit_exact():
    osim = oracle_taylor_green(4)
    pos, ids = lcg_positions(4000)
    # a few particles right at cell faces, box faces and box corners
    h = 1. / 16
    extra = np.array([[0., 0., 0.], [h, -h, 2 * h], [0.5, 0.1, 0.1], [-0.5, -0.5, -0.5],
                      [0.5 - 1e-9, 0.5 - 1e-9, 0.3], [0.49, -0.49, 0.49]])
    pos = np.vstack([pos, extra])
    ids = np.arange(1, len(pos) + 1, dtype=np.uint32)
    opl, gpl = _run(osim, PERIODIC, pos, ids, 6)
    # identical cell indices follow from identical positions; check explicitly on the oracle
    gp, gi = gpl.download()
    op, oi = opl.state()
    for q in range(0, len(gi), 97):
        assert opl.locate(gp[q]) == opl.locate(op[q])


def test_tracers_2d_periodic_bit_exact():
    osim = oracle_reynolds(5)
    pos, ids = lcg_positions(1500, dim=2)
    _run(osim, PERIODIC, pos, ids, 8, end=2.)


@pytest.mark.parametrize("sort_every", [0, 1, 3])
def test_tracers_closed_box_drop_list_identical(sort_every):
    """particles leave through a non-periodic side and are dropped; the storage order (never
    sorted, sorted by cell every event, every third event) must not change positions, ids or
    the order of the downloaded list"""
    level = 4
    side = [O.SIDE_BOUNDARY] * 6
    osim = O.Sim(3, level, side)
    x, y, z = osim.dom.centres()
    osim.u[0].interior()[...] = 0.8 + 0. * x * y * z
    osim.u[1].interior()[...] = 0.1 * np.sin(np.pi * x) + 0. * y * z
    osim.u[2].interior()[...] = 0. * x * y * z
    pos, ids = lcg_positions(2000)
    gd, gs = _device_sim(osim, side)
    # no flow solve here: fixed velocity field with symmetry ghosts, fixed dt
    L = O.lib()
    for c in range(3):
        L.go_bc(osim.u[c].ptr, osim.u[c].ptr, level)
        gd.bc(gs.u[c])
    osim.advection_params.dt = 0.03
    gs.advection_params.dt = 0.03
    opl = O.Particles(osim, pos, ids)
    gpl = gfship.ParticleList(gs, pos, ids)
    gpl.set_sort_interval(sort_every)
    for k in range(10):
        opl.event()
        gpl.event()
        op, oi = opl.state()
        gp, gi = gpl.download()
        assert np.array_equal(oi, gi) and np.array_equal(op, gp), k
    assert 0 < gpl.count() < 2000     # some particles left through the +x side and were dropped


class _MirrorTransport:
    """One box of a 2 x 1 x 1 lattice whose other box is its mirror image (same fields, same
    particles): what the neighbour sends is what this box sends.  Drives the ORACLE as the
    reference of the two-box device run below."""

    def __init__(self, grid):
        import torch
        self.torch, self.grid, self.rank = torch, grid, 0
        self.device = torch.device("cpu")
        self._bufs = {}

    def buffers(self, key, nface):
        from gfship import distributed as D
        import multibox as M
        return D.Transport.buffers(self, key, nface)

    def exchange(self, send_sides, snd, recv_sides, rcv):
        for r in recv_sides:
            rcv[r].copy_(snd[r ^ 1])

    def allreduce(self, vals, op):
        a = np.array(vals, dtype=np.float64)
        return a + a if op == 0 else a

    def exchange_records(self, out, rs=7):
        return {r: out[r ^ 1] for r in out if len(out[r ^ 1])}


def test_particles_migrate_between_two_boxes_on_one_gpu():
    """particles leaving through a GfsBoundaryMpi side are packed on the device, handed to the
    migration hook and join the neighbour's list (modules/particulatecommon.c:3218-3312): two device
    boxes in one process against the oracle run of one box of the same lattice"""
    import threading
    import torch
    from gfship import distributed as D
    import multibox as M
    level, nsteps, npart = 5, 4, 3000
    grid = D.BoxGrid(2, 3)
    pos, ids = lcg_positions(npart)
    pos[:200, 0] = 0.5 - 2e-4 * (1 + np.arange(200))          # a crowd next to the +x MPI side
    pos[200:300, 0] = -0.5 + 2e-4 * (1 + np.arange(100))

    def init(sim_u, x, y, z):
        from flow_cases import taylor_green_3d
        for c, a in enumerate(taylor_green_3d(x, y, z)):
            sim_u[c][...] = a + (0.7 if c == 0 else 0.)          # through-flow along x

    # oracle reference: one box + its mirror image
    osim = O.Sim(3, level, grid.sides(0))
    mt = _MirrorTransport(grid)
    ohooks = M.OracleHooks(O.lib(), osim.dom.ptr, 3, mt)
    init([osim.u[c].interior() for c in range(3)], *osim.dom.centres())
    opl = O.Particles(osim, pos, ids)
    osim.start()
    ostates, moved = [], 0
    for _ in range(nsteps):
        opl.event()
        out = {d: opl.outbox(d) for d in grid.external_sides()}
        moved += sum(len(a) for a in out.values())
        opl.clear_outbox()
        for d, a in sorted(mt.exchange_records(out).items()):
            opl.append(a)
        ostates.append(opl.state())
        osim.step()
    assert moved > 50

    fabric = M.LocalFabric(2)
    dev = torch.device("cuda", 0)
    results, errors = [None, None], []
    u0 = [osim.dom.field() for _ in range(3)]
    init([f.interior() for f in u0], *osim.dom.centres())

    def worker(rank):
        try:
            gd = gfship.Domain(3, level, grid.sides(rank))
            gs = gfship.Simulation(gd)
            tr = M.LocalTransport(grid, rank, fabric, dev)
            hooks = D.DeviceHooks(gd, tr)
            for c in range(3):
                gs.u[c].upload(u0[c].leaf())
            gpl = gfship.ParticleList(gs, pos, ids)
            gpl.set_sort_interval(2)
            mig = D.ParticleMigration(gpl, tr)
            gs.start()
            states = []
            for _ in range(nsteps):
                gpl.event()
                states.append(gpl.download() + (gpl.count(),))
                gs.step()
            gd.synchronize()
            results[rank] = (states, gd, gs, hooks, mig, gpl)
        except Exception as e:       # pragma: no cover
            import traceback
            traceback.print_exc()
            errors.append(e)
            fabric.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert not errors, errors
    for rank in range(2):
        for k, (gp, gi, cnt) in enumerate(results[rank][0]):
            op, oi = ostates[k]
            assert cnt == len(oi) == len(gi), (rank, k)
            a, b = np.argsort(gi, kind="stable"), np.argsort(oi, kind="stable")
            assert np.array_equal(gi[a], oi[b]), (rank, k)
            assert np.array_equal(gp[a], op[b]), (rank, k, np.abs(gp[a] - op[b]).max())
    del ohooks


@pytest.mark.parametrize("particulate", [False, True])
def test_particles_migrate_through_the_library_communicator(particulate):
    """the same hand-over without any hook: with a communicator on the domain (here one rank whose x
    sides face the box itself) the packets travel by ncclSend / ncclRecv, counts first, then the
    records (csrc/transport.hip comm_migrate) -- against the oracle box with the mirror transport"""
    import multibox as M
    level, nsteps, npart = 5, 4, 2500
    side = [gfship.SIDE_EXTERNAL, gfship.SIDE_EXTERNAL] + [gfship.SIDE_PERIODIC] * 4
    pos, ids = lcg_positions(npart)
    pos[:200, 0] = 0.5 - 2e-4 * (1 + np.arange(200))
    pos[200:300, 0] = -0.5 + 2e-4 * (1 + np.arange(100))
    rng = np.random.default_rng(3)
    vel = 0.3 * rng.standard_normal((npart, 3))
    vol = 1e-3 * (0.5 + rng.random(npart))
    mass = vol * (0.5 + 2.5 * rng.random(npart))
    forces = [O.FORCE_DRAG, O.FORCE_BUOY, O.FORCE_INERTIAL]
    nu = 1e-2 if particulate else 0.
    osim = O.Sim(3, level, side)
    x, y, z = osim.dom.centres()
    from flow_cases import taylor_green_3d
    for c, a in enumerate(taylor_green_3d(x, y, z)):
        osim.u[c].interior()[...] = a + (0.7 if c == 0 else 0.)
        if nu:
            osim.set_viscosity(c, nu)
    grid = type("G", (), {"external_sides": lambda s: [0, 1]})()
    mt = _MirrorTransport(grid)
    ohooks = M.OracleHooks(O.lib(), osim.dom.ptr, 3, mt)
    gd, gs = _device_sim(osim, side)
    for c in range(3):
        if nu:
            gs.set_viscosity(c, nu)
    gd.comm_init(gfship.comm_unique_id(), 0, 1, (1, 1, 1))
    opl = O.Particles(osim, pos, ids)
    gpl = gfship.ParticleList(gs, pos, ids)
    gpl.set_sort_interval(2)
    if particulate:
        opl.set_particulate(vel, mass, vol)
        gpl.set_particulate(vel, mass, vol)
        opl.set_forces(forces, (0., -0.4, 0.))
        gpl.set_forces(forces, (0., -0.4, 0.))
    osim.start()
    gs.start()
    moved = 0
    for k in range(nsteps):
        opl.event()
        out = {d: opl.outbox(d) for d in (0, 1)}
        moved += sum(len(a) for a in out.values())
        opl.clear_outbox()
        for d, a in sorted(mt.exchange_records(out, rs=15 if particulate else 7).items()):
            opl.append(a)
        gpl.event()
        op, oi = opl.state()
        gp, gi = gpl.download()
        assert gpl.count() == len(oi) == len(gi), k
        a, b = np.argsort(gi, kind="stable"), np.argsort(oi, kind="stable")
        assert np.array_equal(gi[a], oi[b]), k
        if particulate:
            assert _rel_err(op[b], gp[a]) <= 1e-12, k
            assert _rel_err(opl.particulate_state()[0][b], gpl.particulate_state()[0][a]) <= 1e-12, k
        else:
            assert np.array_equal(gp[a], op[b]), (k, np.abs(gp[a] - op[b]).max())
        osim.step()
        gs.step()
    assert moved > 50
    del ohooks
    gs.destroy()
    gd.destroy()


# ---------------------------------------------------------------------------------------------
# GfsParticulate with forces: modules/particulatecommon.c:91-842
# ---------------------------------------------------------------------------------------------

ALL_FORCES = [O.FORCE_INERTIAL, O.FORCE_ADDEDMASS, O.FORCE_LIFT, O.FORCE_DRAG, O.FORCE_BUOY]


def _rel_err(a, b):
    scale = max(np.abs(a).max(), 1e-300)
    return np.abs(a - b).max() / scale


def _run_particulates(osim, side, pos, ids, vel, mass, vol, forces, gravity, nsteps, nu, sort_every=None,
                      end=None, coefficients=None):
    gd, gs = _device_sim(osim, side)
    if end is not None:
        gs.set_time(end=end)
    for c in range(osim.dim):
        if nu:
            gs.set_viscosity(c, nu)
    osim.start()
    gs.start()
    opl = O.Particles(osim, pos, ids)
    gpl = gfship.ParticleList(gs, pos, ids)
    if sort_every is not None:
        gpl.set_sort_interval(sort_every)
    opl.set_particulate(vel, mass, vol)
    gpl.set_particulate(vel, mass, vol)
    opl.set_forces(forces, gravity)
    gpl.set_forces(forces, gravity)
    for f, (text, fn) in (coefficients or {}).items():
        opl.set_coefficient(f, fn)
        gpl.set_force_coefficient(f, text)
    worst = 0.
    for k in range(nsteps):
        opl.event()
        gpl.event()
        op, oi = opl.state()
        gp, gi = gpl.download()
        # same survivors in the same order; positions, velocities, forces within 1e-12 relative
        # L-infinity (the only operation not bit-reproducible is pow (Re, 0.5) of the drag law,
        # sqrt on the device)
        assert np.array_equal(oi, gi), k
        ov, om, of = opl.particulate_state()
        gv, gm, gf = gpl.particulate_state()
        for a, b, what in ((op, gp, "pos"), (ov, gv, "vel"), (of, gf, "force"), (om, gm, "mass")):
            e = _rel_err(a, b)
            worst = max(worst, e)
            assert e <= 1e-12, (k, what, e)
        # cell indices bit-identical
        for q in range(0, len(gi), 37):
            assert opl.locate(gp[q]) == opl.locate(op[q])
        osim.step()
        gs.step()
    return worst, opl, gpl


def _particulate_case(n, dim, seed):
    rng = np.random.default_rng(seed)
    pos, ids = lcg_positions(n, dim=dim)
    vel = 0.3 * rng.standard_normal((n, 3))
    if dim == 2:
        vel[:, 2] = 0.
    vol = 1e-3 * (0.5 + rng.random(n))
    mass = vol * (0.5 + 2.5 * rng.random(n))
    return pos, ids, vel, mass, vol


@pytest.mark.parametrize("sort_every", [0, 2])
def test_particulates_all_forces_taylor_green_3d(sort_every):
    nu = 1e-2
    osim = oracle_taylor_green(4)
    for c in range(3):
        osim.set_viscosity(c, nu)
    pos, ids, vel, mass, vol = _particulate_case(1500, 3, 11)
    worst, opl, gpl = _run_particulates(osim, PERIODIC, pos, ids, vel, mass, vol, ALL_FORCES,
                                        (0., 0.5, 0.), 6, nu, sort_every=sort_every)
    # the added mass accumulates at every event (:424), on both sides alike
    om = opl.particulate_state()[1]
    idx = np.searchsorted(ids, opl.state()[1])
    assert np.allclose(om, mass[idx] + 6 * 0.5 * vol[idx], rtol=1e-13)
    assert worst <= 1e-12


def test_particulates_with_coefficient_functions():
    """GfsForceCoeff objects carrying a GfsFunction (modules/particulatecommon.c:166-210): the C text is
    compiled for the device with hipRTC and evaluated per particle; the oracle calls the same
    expressions on the host.  Schiller-Naumann drag, a lift coefficient given as a { block }, an
    added-mass coefficient of the diameter and the relative velocity."""
    import math
    nu = 1e-2
    osim = oracle_taylor_green(4)
    for c in range(3):
        osim.set_viscosity(c, nu)
    pos, ids, vel, mass, vol = _particulate_case(1200, 3, 21)
    forces = [O.FORCE_INERTIAL, O.FORCE_ADDEDMASS, O.FORCE_LIFT, O.FORCE_DRAG, O.FORCE_BUOY]
    coefficients = {
        1: ("0.5 + 0.1*Pdia + 0.01*fabs (Wrelp)",
            lambda rep, u, v, w, d: 0.5 + 0.1 * d + 0.01 * abs(w)),
        2: ("{ double a = 0.3 + 0.1*Urelp; if (Rep > 1.) a += 0.05*Vrelp; return a; }",
            lambda rep, u, v, w, d: (0.3 + 0.1 * u) + (0.05 * v if rep > 1. else 0.)),
        3: ("24./Rep*(1. + 0.15*pow (Rep, 0.687))",
            lambda rep, u, v, w, d: 24. / rep * (1. + 0.15 * math.pow(rep, 0.687))),
    }
    worst, opl, gpl = _run_particulates(osim, PERIODIC, pos, ids, vel, mass, vol, forces, (0., 0.5, 0.), 5, nu,
                                        sort_every=2, coefficients=coefficients)
    assert worst <= 1e-12
    # the coefficients are really used: the default laws give another state
    w2, opl2, gpl2 = _run_particulates(_tg_with_nu(nu), PERIODIC, pos, ids, vel, mass, vol, forces,
                                       (0., 0.5, 0.), 5, nu, sort_every=2)
    v1, v2 = gpl.particulate_state()[0], gpl2.particulate_state()[0]
    assert len(v1) != len(v2) or _rel_err(v1, v2) > 1e-6


def _tg_with_nu(nu):
    osim = oracle_taylor_green(4)
    for c in range(3):
        osim.set_viscosity(c, nu)
    return osim


def test_a_coefficient_function_that_does_not_compile_is_reported():
    osim = oracle_taylor_green(3)
    gd, gs = _device_sim(osim, PERIODIC)
    pos, ids, vel, mass, vol = _particulate_case(10, 3, 5)
    gpl = gfship.ParticleList(gs, pos, ids)
    gpl.set_particulate(vel, mass, vol)
    gpl.set_forces([O.FORCE_DRAG], (0., 0., 0.))
    with pytest.raises(gfship.GfshipError, match="does not compile"):
        gpl.set_force_coefficient(0, "24./Rep +* nonsense(")
    with pytest.raises(gfship.GfshipError, match="coefficient"):
        gpl.set_forces([O.FORCE_BUOY], (0., 0., 0.)) or gpl.set_force_coefficient(0, "1.")


def test_particulates_2d_periodic():
    nu = 1e-3
    osim = oracle_reynolds(5)
    for c in range(2):
        osim.set_viscosity(c, nu)
    pos, ids, vel, mass, vol = _particulate_case(800, 2, 12)
    _run_particulates(osim, PERIODIC, pos, ids, vel, mass, vol,
                      [O.FORCE_DRAG, O.FORCE_LIFT, O.FORCE_INERTIAL], (0., 0., 0.), 6, nu, end=2.)


def test_particulates_sediment_out_of_a_closed_box():
    """lid-driven cavity: heavy particulates fall through the bottom wall and leave the list (no
    periodic side puts them back); drag + buoyancy only"""
    from flow_cases import oracle_lid
    osim = oracle_lid(level=5, nu=1e-2)
    n = 300
    pos, ids, vel, mass, vol = _particulate_case(n, 2, 13)
    pos[:, 1] = -0.5 + 0.1 * (pos[:, 1] + 0.5)          # near the bottom wall
    vel[...] = 0.
    mass = 5. * vol
    from test_gpu_timestep import _device_lid
    gd, gs = _device_lid(osim, 5, 1e-2)
    gs.set_time(end=300.)
    osim.start()
    gs.start()
    opl = O.Particles(osim, pos, ids)
    gpl = gfship.ParticleList(gs, pos, ids)
    for pl in (opl, gpl):
        pl.set_particulate(vel, mass, vol)
        pl.set_forces([O.FORCE_DRAG, O.FORCE_BUOY], (0., -20., 0.))
    counts = []
    for k in range(12):
        opl.event()
        gpl.event()
        op, oi = opl.state()
        gp, gi = gpl.download()
        assert np.array_equal(oi, gi), k
        if len(oi):
            assert _rel_err(op, gp) <= 1e-12
            assert _rel_err(opl.particulate_state()[0], gpl.particulate_state()[0]) <= 1e-12
        counts.append(len(oi))
        osim.step()
        gs.step()
    assert counts[-1] < counts[0]


def test_particulates_migrate_between_two_boxes_on_one_gpu():
    """particulates crossing a GfsBoundaryMpi side carry their velocity, mass, volume and force with
    them (15-double records): two device boxes in one process against the oracle run of one box
    of the same lattice and its mirror image; drag + lift + buoyancy in a viscous through-flow"""
    import threading
    import torch
    from gfship import distributed as D
    import multibox as M
    level, nsteps, npart, nu = 4, 5, 1200, 1e-2
    grid = D.BoxGrid(2, 3)
    pos, ids, vel, mass, vol = _particulate_case(npart, 3, 21)
    pos[:150, 0] = 0.5 - 3e-4 * (1 + np.arange(150))          # a crowd next to the +x MPI side
    vel[:150, 0] = 0.6
    forces, gravity = [O.FORCE_DRAG, O.FORCE_LIFT, O.FORCE_BUOY], (0., 0.3, 0.)

    def init(sim_u, x, y, z):
        from flow_cases import taylor_green_3d
        for c, a in enumerate(taylor_green_3d(x, y, z)):
            sim_u[c][...] = a + (0.7 if c == 0 else 0.)

    osim = O.Sim(3, level, grid.sides(0))
    for c in range(3):
        osim.set_viscosity(c, nu)
    mt = _MirrorTransport(grid)
    ohooks = M.OracleHooks(O.lib(), osim.dom.ptr, 3, mt)
    init([osim.u[c].interior() for c in range(3)], *osim.dom.centres())
    opl = O.Particles(osim, pos, ids)
    opl.set_particulate(vel, mass, vol)
    opl.set_forces(forces, gravity)
    osim.start()
    ostates, moved = [], 0
    for _ in range(nsteps):
        opl.event()
        out = {d: opl.outbox(d) for d in grid.external_sides()}
        moved += sum(len(a) for a in out.values())
        assert all(a.shape[1] == 15 for a in out.values())
        opl.clear_outbox()
        for d, a in sorted(mt.exchange_records(out).items()):
            opl.append(a)
        ostates.append(opl.state() + opl.particulate_state())
        osim.step()
    assert moved > 30

    fabric = M.LocalFabric(2)
    dev = torch.device("cuda", 0)
    results, errors = [None, None], []
    u0 = [osim.dom.field() for _ in range(3)]
    init([f.interior() for f in u0], *osim.dom.centres())

    def worker(rank):
        try:
            gd = gfship.Domain(3, level, grid.sides(rank))
            gs = gfship.Simulation(gd)
            for c in range(3):
                gs.set_viscosity(c, nu)
            tr = M.LocalTransport(grid, rank, fabric, dev)
            hooks = D.DeviceHooks(gd, tr)
            for c in range(3):
                gs.u[c].upload(u0[c].leaf())
            gpl = gfship.ParticleList(gs, pos, ids)
            gpl.set_sort_interval(2)
            gpl.set_particulate(vel, mass, vol)
            gpl.set_forces(forces, gravity)
            mig = D.ParticleMigration(gpl, tr)
            gs.start()
            states = []
            for _ in range(nsteps):
                gpl.event()
                states.append(gpl.download() + gpl.particulate_state())
                gs.step()
            gd.synchronize()
            results[rank] = (states, gd, gs, hooks, mig, gpl)
        except Exception as e:       # pragma: no cover
            import traceback
            traceback.print_exc()
            errors.append(e)
            fabric.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert not errors, errors
    for rank in range(2):
        for k, (gp, gi, gv, gm, gf) in enumerate(results[rank][0]):
            op, oi, ov, om, of = ostates[k]
            assert len(oi) == len(gi), (rank, k)
            a, b = np.argsort(gi, kind="stable"), np.argsort(oi, kind="stable")
            assert np.array_equal(gi[a], oi[b]), (rank, k)
            for x, y, what in ((gp, op, "pos"), (gv, ov, "vel"), (gm, om, "mass"), (gf, of, "force")):
                assert _rel_err(y[b], x[a]) <= 1e-12, (rank, k, what)
