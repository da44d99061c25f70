"""GPU parity of the projection + advection time step (libgfship through the C ABI) against the
CPU oracle: bit-exact fields, face velocities and time steps, plus the reference's own golden
divergence history (test/reynolds/div5.ref) through the device path."""
import os

import numpy as np
import pytest

import gfship
from flow_cases import (PERIODIC, oracle_periodic, oracle_reynolds, oracle_taylor_green,
                        periodic_init, reynolds_init, taylor_green_3d)
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _with_ghosts(a, dim):
    n = a.shape[0]
    b = np.zeros((n + 2,) * dim)
    b[(slice(1, -1),) * dim] = a
    return b


def _interior(a, dim):
    return a[(slice(1, -1),) * dim]


def _device_sim(osim, side):
    """a device simulation initialised from the oracle simulation's current state"""
    dim, depth = osim.dim, osim.depth
    gd = gfship.Domain(dim, depth, side)
    gs = gfship.Simulation(gd)
    for c in range(dim):
        gs.u[c].upload(osim.u[c].leaf())
    for name in ("projection_params", "approx_projection_params"):
        op, gp = getattr(osim, name), getattr(gs, name)
        for f in ("tolerance", "nrelax", "erelax", "minlevel", "nitermax", "nitermin", "omega"):
            setattr(gp, f, getattr(op, f))
    gs.advection_params.cfl = osim.advection_params.cfl
    gs.advection_params.gradient = osim.advection_params.gradient
    return gd, gs


def _assert_same_state(osim, gs, what=""):
    dim = osim.dim
    for c in range(dim):
        assert np.array_equal(osim.u[c].interior(), _interior(gs.u[c].download(), dim)), \
            "%s U[%d]" % (what, c)
        assert np.array_equal(osim.g[c].interior(), _interior(gs.g[c].download(), dim)), \
            "%s g[%d]" % (what, c)
    assert np.array_equal(osim.p.interior(), _interior(gs.p.download(), dim)), what + " P"
    assert np.array_equal(osim.pmac.interior(), _interior(gs.pmac.download(), dim)), what + " Pmac"
    assert osim.dt == gs.dt, what + " dt"
    assert osim.t == gs.t


def _assert_same_un(osim, gs, what=""):
    """MAC velocities: oracle keeps cell.f[2c].un, the device the + face of each cell"""
    dim = osim.dim
    n = 1 << osim.depth
    for c in range(dim):
        a, b = osim.un(2 * c), gs.un(c)
        # faces with tangential coordinates in 1..n and normal coordinate in 0..n
        sl = [slice(1, n + 1)] * dim
        sl[dim - 1 - c] = slice(0, n + 1)
        assert np.array_equal(a[tuple(sl)], b[tuple(sl)]), "%s un[%d]" % (what, c)


@pytest.mark.parametrize("level", [4, 5])
def test_reynolds_2d_steps_bit_exact(level):
    osim = oracle_reynolds(level)
    gd, gs = _device_sim(osim, PERIODIC)
    gs.set_time(end=2.)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    _assert_same_un(osim, gs, "start")
    for k in range(6):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)
        assert gs.projection_params.niter == osim.projection_params.niter
        assert gs.approx_projection_params.residual.infty == \
            osim.approx_projection_params.residual.infty


def test_reynolds_div5_golden_through_device(golden_dir):
    """every printed digit of the reference's test/reynolds/div5.ref (80 steps)"""
    ref = [l.split() for l in open(os.path.join(golden_dir, "reference", "reynolds_div5.ref"))]
    osim = oracle_reynolds(5)
    gd, gs = _device_sim(osim, PERIODIC)
    gs.set_time(end=2.)
    gs.start()
    k = 0
    while gs.t < 2.:
        nm = gs.divergence_norm()
        got = ("Divergence time: %g first: % 10.3e second: % 10.3e infty: % 10.3e"
               % (gs.t, nm.first, nm.second, nm.infty)).split()
        if k > 0:
            assert got == ref[k], (k, got, ref[k])
        gs.step()
        k += 1
    assert k == 80


def test_periodic_2d_translating_vortex_bit_exact():
    osim = oracle_periodic(5)
    gd, gs = _device_sim(osim, PERIODIC)
    gs.set_time(end=0.5)
    osim.start()
    gs.start()
    for k in range(5):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)


@pytest.mark.parametrize("level", [3, 4, 5])
def test_taylor_green_3d_steps_bit_exact(level):
    """SURVEY.md 8d config C (triply periodic Taylor-Green, default parameters) at a size the
    oracle finishes in seconds"""
    osim = oracle_taylor_green(level)
    gd, gs = _device_sim(osim, PERIODIC)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    _assert_same_un(osim, gs, "start")
    for k in range(3):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)
    assert gs.cfl() == O.lib().go_domain_cfl(osim.ptr)
    on, gn = osim.divergence_norm(), gs.divergence_norm()
    assert gn.infty == on.infty
    assert gn.second == pytest.approx(on.second, rel=1e-12)
    # gfs_cell_coarse_init (simulation.c:530-533): the non-leaf levels hold the restriction of the
    # state before the approximate projection of the step, exactly as the reference's loop leaves them
    for l in (level - 1, 2, 0):
        for name, of, gf in (("U", osim.u[0], gs.u[0]), ("P", osim.p, gs.p)):
            assert np.array_equal(_interior(of.level(l), 3), _interior(gf.download(l), 3)), (name, l)


@pytest.mark.parametrize("dim", [2, 3])
def test_symmetry_box_steps_bit_exact(dim):
    """closed box (default symmetry BCs on every side): exercises the non-periodic ghost and
    face-BC paths and the boundary-deferred flux order"""
    level = 4
    side = [O.SIDE_BOUNDARY] * 6
    osim = O.Sim(dim, level, side)
    cs = osim.dom.centres()
    rng = np.random.default_rng(2)
    if dim == 2:
        x, y = cs
        vel = [np.sin(np.pi * (x + .5)) * np.cos(np.pi * (y + .5)),
               -np.cos(np.pi * (x + .5)) * np.sin(np.pi * (y + .5))]
    else:
        x, y, z = cs
        vel = [np.sin(np.pi * (x + .5)) * np.cos(np.pi * (y + .5)) * np.cos(np.pi * (z + .5)),
               -np.cos(np.pi * (x + .5)) * np.sin(np.pi * (y + .5)) * np.cos(np.pi * (z + .5)),
               0. * x * y * z]
    for c in range(dim):
        osim.u[c].interior()[...] = vel[c] + 0.01 * rng.standard_normal(vel[c].shape)
    gd, gs = _device_sim(osim, side)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(3):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)


def test_tracer_advection_bit_exact_and_conservative():
    """GfsVariableTracer: van Leer + gfs_face_advection_flux; the sum of T is conserved to
    round-off on a periodic box (the invariant of test/conservation)."""
    level = 4
    osim = oracle_taylor_green(level)
    ot = osim.add_tracer()
    x, y, z = osim.dom.centres()
    T0 = np.exp(-40. * (x ** 2 + y ** 2 + z ** 2))
    ot.interior()[...] = T0
    gd, gs = _device_sim(osim, PERIODIC)
    gt = gs.add_tracer()
    gt.upload(ot.leaf())
    osim.start()
    gs.start()
    for k in range(3):
        osim.step()
        gs.step()
        assert np.array_equal(ot.interior(), _interior(gt.download(), 3)), k
    assert abs(_interior(gt.download(), 3).sum() - T0.sum()) < 1e-10 * T0.sum()


def test_two_boxes_with_mpi_sides_on_one_gpu_reproduce_single_box():
    """GFSHIP_SIDE_EXTERNAL path of libgfship (halo pack/unpack kernels, exchange and reduce hooks,
    external ghost streams of the pipelined sweep): two boxes in one process, each holding one
    period of the Taylor-Green field, exchanged by the in-process transport.  Every box must equal
    the single periodic box bit for bit (the MPI-lagged ghosts equal the periodic ghosts)."""
    import threading
    import torch
    from gfship import distributed as D
    import multibox as M
    level, nsteps = 5, 2
    osim = oracle_taylor_green(level)
    grid = D.BoxGrid(2, 3)
    fabric = M.LocalFabric(2)
    dev = torch.device("cuda", 0)
    sims, errors = [None, None], []

    def worker(rank):
        try:
            gd = gfship.Domain(3, level, grid.sides(rank))
            gs = gfship.Simulation(gd)
            hooks = D.DeviceHooks(gd, M.LocalTransport(grid, rank, fabric, dev))
            for c in range(3):
                gs.u[c].upload(osim.u[c].leaf())
            gs.start()
            for _ in range(nsteps):
                gs.step()
            gd.synchronize()
            sims[rank] = (gd, gs, hooks)
        except Exception as e:       # pragma: no cover
            errors.append(e)
            fabric.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    osim.start()
    for _ in range(nsteps):
        osim.step()
    for rank in range(2):
        gd, gs, hooks = sims[rank]
        _assert_same_state(osim, gs, "box %d" % rank)


# ---------------------------------------------------------------------------------------------
# implicit viscosity (GfsSourceDiffusion) and the lid-driven cavity, BASELINE config A
# ---------------------------------------------------------------------------------------------

def _device_lid(osim, level, nu):
    side = [O.SIDE_BOUNDARY] * 6
    gd, gs = _device_sim(osim, side)
    n = 1 << level
    nface = n ** (osim.dim - 1)
    for c in range(osim.dim):
        for d in range(2 * osim.dim):
            val = np.full(nface, 1. if (c == 0 and d == 2) else 0.)
            gs.u[c].set_bc(d, gfship.BC_DIRICHLET, val)
        gs.set_viscosity(c, nu)
    return gd, gs


def _assert_same_diffusion(osim, gs, what):
    for c in range(osim.dim):
        a, b = osim.diffusion_params(c), gs.diffusion_params(c)
        assert a.niter == b.niter, what
        assert a.residual.infty == b.residual.infty, what
        assert a.residual_before.infty == b.residual_before.infty, what


def test_lid_2d_steps_bit_exact():
    """test/lid/lid.gfs on 64^2: Dirichlet walls, implicit viscosity (diffusion multigrid, MAC
    source term in the predictor, acceleration term of the CFL condition)"""
    from flow_cases import oracle_lid
    osim = oracle_lid(6, 1e-3)
    gd, gs = _device_lid(osim, 6, 1e-3)
    gs.set_time(end=300.)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(25):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)
        _assert_same_diffusion(osim, gs, "step %d" % k)


@pytest.mark.parametrize("level", [3, 5])
def test_viscous_box_3d_steps_bit_exact(level):
    """3-D lid-driven box with implicit viscosity: level 3 runs the LDS relax loop, level 5 the
    per-hyperplane launches, both with the diffusion cell update"""
    side = [O.SIDE_BOUNDARY] * 6
    osim = O.Sim(3, level, side)
    n = 1 << level
    for c in range(3):
        for d in range(6):
            osim.u[c].set_bc(d, O.BC_DIRICHLET, np.full(n * n, 1. if (c == 0 and d == 2) else 0.))
        osim.set_viscosity(c, 1e-2)
    gd, gs = _device_lid(osim, level, 1e-2)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(3):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_diffusion(osim, gs, "step %d" % k)


@pytest.mark.parametrize("dim,beta", [(2, 0.5), (3, 0.5), (2, 1.)])
def test_diffusion_solver_matches_oracle(dim, beta):
    """gfs_diffusion_coefficients / rhs / gfs_diffusion on their own, Crank-Nicholson
    (beta = 0.5) included, on a periodic box with a random field"""
    level = 4
    osim = O.Sim(dim, level, PERIODIC)
    rng = np.random.default_rng(11)
    n = 1 << level
    v0 = rng.standard_normal((n,) * dim)
    ov, orhs = osim.dom.field(), osim.dom.field()
    ov.interior()[...] = v0
    orhs.interior()[...] = v0
    O.lib().go_bc(ov.ptr, ov.ptr, level)
    par = osim.diffusion_params(0)
    par.beta = beta
    osim.variable_diffusion(ov, orhs, 0.05, 0.1, par)

    gd = gfship.Domain(dim, level, PERIODIC)
    gv, grhs, grhoc = gd.variable(), gd.variable(), gd.variable()
    gv.upload(_with_ghosts(v0, dim))
    grhs.upload(_with_ghosts(v0, dim))
    gd.bc(gv, gv, level)
    gpar = gd.params()
    gpar.tolerance = 1e-6
    gpar.beta = beta
    gd.diffusion_coefficients(0.05, 0.1, grhoc, beta)
    gd.diffusion_rhs(gv, grhs, grhoc, beta)
    gd.diffusion(gpar, gv, grhs, grhoc)
    assert gpar.niter == par.niter and gpar.niter >= 1
    assert gpar.residual.infty == par.residual.infty
    assert np.array_equal(ov.interior(), _interior(gv.download(), dim))
    assert np.array_equal(orhs.interior(), _interior(grhs.download(), dim))


def test_lid_ghia_through_device(golden_dir):
    """test/lid end to end on the device: run to the GfsEventStop steady state (U changes by
    less than 1e-4 over 10 steps), same number of steps and bit-identical fields as the oracle,
    then the Ghia et al. profile check of test/lid/lid.sh:11-12 on the device fields."""
    from flow_cases import oracle_lid, run_until_steady
    osim = oracle_lid(6, 1e-3)
    gd, gs = _device_lid(osim, 6, 1e-3)
    gs.set_time(end=300.)
    ohist = run_until_steady(osim, osim.u[0])

    class _U:   # the EventStop driver reads var.interior()
        def interior(self):
            return _interior(gs.u[0].download(), 2)
    ghist = run_until_steady(gs, _U())
    assert [(i, t) for i, t, _ in ghist] == [(i, t) for i, t, _ in ohist]
    assert ghist[-1][2] == ohist[-1][2] <= 1e-4
    _assert_same_state(osim, gs, "steady")
    # profiles from the device fields (ghost layer included), interpolated like OutputLocation
    chk = O.Sim(2, 6, [O.SIDE_BOUNDARY] * 6)
    for c in range(2):
        chk.u[c].leaf()[...] = gs.u[c].download()
    g = os.path.join(golden_dir, "reference")
    xp = chk.output_location(np.loadtxt(os.path.join(g, "lid_xprofile")), [chk.u[0], chk.u[1]])
    yp = chk.output_location(np.loadtxt(os.path.join(g, "lid_yprofile")), [chk.u[0], chk.u[1]])
    gx = np.loadtxt(os.path.join(g, "xprof.ghia"))
    gy = np.loadtxt(os.path.join(g, "yprof.ghia"))
    ex = np.abs(np.interp(gx[:, 0], xp[:, 1], xp[:, 3]) - gx[:, 1]).max()
    ey = np.abs(np.interp(gy[:, 0], yp[:, 0], yp[:, 4]) - gy[:, 1]).max()
    assert ex <= 2e-2 and ey <= 1.7e-2, (ex, ey)


@pytest.mark.parametrize("dim,level", [(2, 5), (3, 5)])
def test_fused_periodic_paths_equal_general_paths(dim, level):
    """periodic boxes take the fused kernels (relax loop in one launch, Godunov step without the
    face-value arrays); RELAX_EXACT_PER_SWEEP selects the general kernels: same bits, with a
    tracer and implicit viscosity in the mix"""
    osim = oracle_reynolds(level) if dim == 2 else oracle_taylor_green(level)
    res = []
    for mode in (gfship.RELAX_EXACT, gfship.RELAX_EXACT_PER_SWEEP):
        gd, gs = _device_sim(osim, PERIODIC)
        gd.set_relax_mode(mode)
        t = gs.add_tracer()
        rng = np.random.default_rng(5)
        t.upload(_with_ghosts(rng.random(((1 << level),) * dim), dim))
        gs.set_viscosity(0, 1e-3)
        gs.start()
        for _ in range(3):
            gs.step()
        res.append([gs.u[c].download() for c in range(dim)] + [gs.p.download(), t.download()] +
                   [gs.un(c) for c in range(dim)] + [np.array([gs.t, gs.dt])])
        gd.destroy()
    for a, b in zip(*res):
        assert np.array_equal(_interior(a, dim) if a.ndim == dim else a,
                              _interior(b, dim) if b.ndim == dim else b)


@pytest.mark.parametrize("gradient,source,lazy", [(0, 0., True), (1, 0., True), (0, -0.7, True), (1, 0., False)])
def test_sweep_kernels_64_several_steps_vs_oracle(gradient, source, lazy, monkeypatch):
    """the Godunov kernels as sweeps along z (advect3_sweep_kernel with the centred correction and the
    first coarse level fused, predict_un_sweep_kernel with the divergence of the MAC projection fused)
    at the smallest size they run on (64^3: one z chunk), over several steps (the first step advects
    with gmac in place of g), with the centred and the van Leer gradient and with a GfsSource; the MAC
    velocities the approximate projection leaves unstored are rebuilt when they are downloaded
    (materialize_un) -- every field, un included, against the oracle after every step"""
    if not lazy:
        monkeypatch.setenv("GFSHIP_NO_LAZY_UN", "1")
    level = 6
    osim = oracle_taylor_green(level)
    osim.u[0].interior()[...] += 0.35         # upwind directions of both signs, faces of zero velocity
    osim.u[2].interior()[...] -= 0.2
    osim.advection_params.gradient = gradient
    gd, gs = _device_sim(osim, PERIODIC)
    if source:
        osim.set_source(1, source)
        gs.set_source(1, source)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(3):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)
        # the level below the leaves (gfs_cell_coarse_init inside the step: filled by the advection sweep)
        for c in range(3):
            assert np.array_equal(osim.u[c].level(level - 1)[1:-1, 1:-1, 1:-1],
                                  gs.u[c].download(level - 1)[1:-1, 1:-1, 1:-1]), (k, c)
    assert gs.cfl() == O.lib().go_domain_cfl(osim.ptr)
    gd.destroy()


# ---------------------------------------------------------------------------------------------
# GfsAdvection with a GfsVariableStreamFunction (test/advection): tracer in solid rotation
# ---------------------------------------------------------------------------------------------

def _blob(x, y):
    r2 = x * x + y * y
    coeff = 20. + 20000. * r2 * r2 * r2 * r2
    return (1. + np.cos(20. * x) * np.cos(20. * y)) * np.exp(-coeff * r2) / 2.


@pytest.mark.parametrize("level,gradient", [(5, 0), (6, 0), (5, 1)])
def test_advection_run_stream_function_bit_exact(level, gradient, golden_dir):
    """advection_run (src/simulation.c:2061-2116): MAC velocities from the stream function
    -4 (x^2 + y^2), one revolution of the tracer blob of test/advection; the device against the
    oracle bit for bit at every step, and (centred gradient, 64^2) the error norms of error.ref"""
    import os
    side = [O.SIDE_BOUNDARY] * 6
    osim = O.Sim(2, level, side)
    T = osim.add_tracer(gradient=gradient)
    x, y = osim.dom.centres()
    T.interior()[...] = _blob(x, y)
    O.lib().go_bc(T.ptr, T.ptr, level)
    osim.stream_function(lambda X, Y: -4. * (X * X + Y * Y))
    osim.set_time(end=0.785398)

    gd = gfship.Domain(2, level, side)
    gs = gfship.Simulation(gd)
    gs.set_time(end=0.785398)
    gT = gs.add_tracer(gradient=gradient)
    gT.upload(T.leaf())
    for c in range(2):
        gs.u[c].upload(osim.u[c].leaf())
        # + face of every cell; ghost cell 0 holds the - face of the first cell
        a = osim.un(2 * c).copy()
        lo = osim.un(2 * c + 1)
        if c == 0:
            a[:, 0] = lo[:, 1]
        else:
            a[0, :] = lo[1, :]
        gs.mac_velocity(c).upload(a)
    k = 0
    while osim.t < 0.785398:
        osim.advection_step()
        gs.advection_step()
        assert osim.dt == gs.dt and osim.t == gs.t, k
        if k % 16 == 0:
            assert np.array_equal(T.interior(), _interior(gT.download(), 2)), k
        k += 1
    assert np.array_equal(T.interior(), _interior(gT.download(), 2))
    if gradient == 0 and level == 6:
        e = _interior(gT.download(), 2) - _blob(x, y)
        rows = [l.split() for l in open(os.path.join(golden_dir, "reference", "advection_error.ref"))]
        ref = {int(r[0]): [float(v) for v in r[1:]] for r in rows}[level]
        assert np.allclose([np.sqrt((e * e).mean()), np.abs(e).max()], ref[1:], rtol=2e-3)


# ---------------------------------------------------------------------------------------------
# GfsSource with a constant intensity on the velocity components (src/source.c:362-500)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("level,nu", [(4, 0.), (5, 0.), (4, 1e-2)])
def test_body_force_taylor_green_3d_bit_exact(level, nu):
    """`Source U / V / W g' on the triply periodic box: the MAC source of the face values (the tiled
    kernels at 32^3, the three-component kernel without viscosity, the one-component kernels with
    it), the centred source, the acceleration time scale of the CFL condition; device against oracle
    bit for bit, and the flow really accelerates"""
    osim = oracle_taylor_green(level)
    g = (0.7, -1.3, 0.4)
    for c in range(3):
        osim.set_source(c, g[c])
        if nu:
            osim.set_viscosity(c, nu)
    gd, gs = _device_sim(osim, PERIODIC)
    for c in range(3):
        gs.set_source(c, g[c])
        if nu:
            gs.set_viscosity(c, nu)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    assert gs.cfl() == O.lib().go_domain_cfl(osim.ptr)
    for k in range(3):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)
    assert gs.cfl() == O.lib().go_domain_cfl(osim.ptr)
    assert abs(osim.u[1].interior().mean() - g[1] * osim.t) < 0.05 * abs(g[1] * osim.t)


@pytest.mark.parametrize("level", [3, 5])
def test_poiseuille_channel_steps_bit_exact(level):
    """the set-up of test/poiseuille (periodic channel, Dirichlet walls, Source U 1, Source V 1,
    SourceViscosity 1 { beta = 1 }): the general (face-value array) kernels with sources, 2-D"""
    side = [O.SIDE_PERIODIC, O.SIDE_PERIODIC, O.SIDE_BOUNDARY, O.SIDE_BOUNDARY, O.SIDE_BOUNDARY, O.SIDE_BOUNDARY]
    osim = O.Sim(2, level, side)
    n = 1 << level
    gd = gfship.Domain(2, level, side)
    gs = gfship.Simulation(gd)
    for d in (2, 3):
        osim.u[0].set_bc(d, O.BC_DIRICHLET, np.zeros(n))
        gs.u[0].set_bc(d, gfship.BC_DIRICHLET, np.zeros(n))
    for c in range(2):
        osim.set_viscosity(c, 1.)
        osim.diffusion_params(c).beta = 1.
        osim.set_source(c, 1.)
        gs.set_viscosity(c, 1.)
        gs.diffusion_params(c).beta = 1.
        gs.set_source(c, 1.)
    for sim in (osim, gs):
        sim.projection_params.tolerance = 1e-6
        sim.approx_projection_params.tolerance = 1e-6
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(6):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
    assert gs.dt == osim.dt and osim.u[0].interior().max() > 0.05


# ---------------------------------------------------------------------------------------------
# GfsPhysicalParams { alpha = ... } (variable density): both projections with gfs_poisson_coefficients
# (alpha), the multigrid with the face weights of every level, gfs_correct_normal_velocities with
# gfs_face_weighted_gradient's weights (src/timestep.c:118-144,306-322,356-444).  The reference holds
# no golden file for a variable-density run without VOF (test/ uses alpha only with interfaces): parity
# is oracle against device, the oracle's weighted multigrid being the one pinned by K8's tests.
# ---------------------------------------------------------------------------------------------

def _alpha_faces(osim, gd, kind):
    """1/rho at the face centres, rho = 1 + 0.5 sin (2 pi x) cos (2 pi y) [cos (2 pi z)]: oracle Fields and
    device Variables in the layout of gfs_poisson_coefficients' alpha (the entry of a cell is its + face
    along c, the ghost entry in front of the first cell its - face)"""
    dim, depth = osim.dim, osim.depth
    n = 1 << depth
    h = 1. / n
    idx = np.arange(n + 2)
    centre = (idx - 0.5) * h - 0.5          # centre of cell i (ghosts included), box [-0.5, 0.5]
    oa, ga = [], []
    for c in range(dim):
        axes = []
        for ax in range(dim):               # array axes are [k, j, i] / [j, i]
            comp = dim - 1 - ax
            axes.append(centre + (0.5 * h if comp == c else 0.))
        grids = np.meshgrid(*axes, indexing="ij")
        xyz = [grids[dim - 1 - comp] for comp in range(dim)]
        rho = 1. + 0.5 * np.sin(2. * np.pi * xyz[0]) * np.cos(2. * np.pi * xyz[1])
        if dim == 3:
            rho = rho * np.cos(2. * np.pi * xyz[2]) + (1. - np.cos(2. * np.pi * xyz[2]))
        a = 1. / rho
        if kind == "periodic":
            ax = dim - 1 - c
            sl0, sln = [slice(None)] * dim, [slice(None)] * dim
            sl0[ax], sln[ax] = 0, -2
            a[tuple(sl0)] = a[tuple(sln)]
        of, gf = O.Field(osim.dom, -1), gd.variable()
        of.leaf()[...] = a
        gf.upload(a)
        oa.append(of)
        ga.append(gf)
    return oa, ga


@pytest.mark.parametrize("dim,level,kind", [(2, 5, "periodic"), (3, 4, "periodic"), (3, 5, "periodic"),
                                            (2, 5, "box"), (3, 5, "box")])
def test_variable_density_steps_bit_exact(dim, level, kind):
    """time steps with GfsPhysicalParams { alpha }: U, g, P, Pmac, the MAC velocities and dt equal to the
    oracle's bits at the start and after every step (32^3: the weighted sweeps on the pipelined kernel)"""
    side = PERIODIC if kind == "periodic" else [O.SIDE_BOUNDARY] * 6
    osim = O.Sim(dim, level, side)
    cs = osim.dom.centres()
    rng = np.random.default_rng(5)
    if kind == "periodic":
        if dim == 2:
            vel = reynolds_init(*cs)
        else:
            vel = taylor_green_3d(*cs)
    else:
        if dim == 2:
            x, y = cs
            vel = [np.sin(np.pi * (x + .5)) * np.cos(np.pi * (y + .5)),
                   -np.cos(np.pi * (x + .5)) * np.sin(np.pi * (y + .5))]
        else:
            x, y, z = cs
            vel = [np.sin(np.pi * (x + .5)) * np.cos(np.pi * (y + .5)) * np.cos(np.pi * (z + .5)),
                   -np.cos(np.pi * (x + .5)) * np.sin(np.pi * (y + .5)) * np.cos(np.pi * (z + .5)),
                   0. * x * y * z]
    for c in range(dim):
        osim.u[c].interior()[...] = vel[c] + 0.01 * rng.standard_normal(np.shape(vel[c]))
    gd, gs = _device_sim(osim, side)
    oa, ga = _alpha_faces(osim, gd, kind)
    osim.set_alpha(oa)
    gs.set_alpha(ga)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    _assert_same_un(osim, gs, "start")
    for k in range(3):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)
    # the weights made a difference: the same run with alpha = NULL gives another pressure
    osim1 = O.Sim(dim, level, side)
    for c in range(dim):
        osim1.u[c].interior()[...] = vel[c] + 0.01 * np.random.default_rng(5).standard_normal(np.shape(vel[c]))
    osim1.start()
    assert not np.array_equal(osim1.p.interior(), osim.p.interior())
    # and back to alpha = NULL on the same device simulation: the unit-weight kernels again
    osim.set_alpha(None)
    gs.set_alpha(None)
    osim.step()
    gs.step()
    _assert_same_state(osim, gs, "alpha = NULL again")


# ---------------------------------------------------------------------------------------------
# the limited gradients of src/fluid.c:563-690 (minmod, superbee, sweby) in the Godunov face values:
# GfsAdvectionParams { gradient = } and GfsVariableTracer { gradient = }
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("gradient", [2, 3, 4])
@pytest.mark.parametrize("dim,kind", [(3, "periodic"), (2, "box")])
def test_limited_gradients_steps_bit_exact(dim, kind, gradient):
    """steps with gfs_center_minmod_gradient / superbee / sweby on the velocity and on a tracer: state,
    MAC velocities and the tracer equal to the oracle's bits"""
    level = 4
    side = PERIODIC if kind == "periodic" else [O.SIDE_BOUNDARY] * 6
    osim = O.Sim(dim, level, side)
    cs = osim.dom.centres()
    rng = np.random.default_rng(40 + gradient)
    if dim == 3:
        vel = taylor_green_3d(*cs)
    else:
        x, y = cs
        vel = [np.sin(np.pi * (x + .5)) * np.cos(np.pi * (y + .5)),
               -np.cos(np.pi * (x + .5)) * np.sin(np.pi * (y + .5))]
    for c in range(dim):
        osim.u[c].interior()[...] = vel[c] + 0.05 * rng.standard_normal(np.shape(vel[c]))
    osim.advection_params.gradient = gradient
    ot = osim.add_tracer(gradient=gradient)
    r2 = sum((q - 0.1) ** 2 for q in cs)
    ot.interior()[...] = (r2 < 0.08) * 1.          # a step profile: every branch of the limiters
    gd, gs = _device_sim(osim, side)
    gt = gs.add_tracer(gradient=gradient)
    gt.upload(_with_ghosts(ot.interior(), dim))
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    for k in range(3):
        osim.step()
        gs.step()
        _assert_same_state(osim, gs, "step %d" % k)
        _assert_same_un(osim, gs, "step %d" % k)
        assert np.array_equal(ot.interior(), _interior(gt.download(), dim)), "tracer, step %d" % k
    # not the van Leer result
    osim1 = O.Sim(dim, level, side)
    for c in range(dim):
        osim1.u[c].interior()[...] = vel[c] + 0.05 * np.random.default_rng(40 + gradient).standard_normal(np.shape(vel[c]))
    osim1.advection_params.gradient = 1
    osim1.start()
    osim1.step()
    assert osim1.dt > 0.
