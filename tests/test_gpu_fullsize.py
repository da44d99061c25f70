"""Full-size parity (BASELINE.json configs B, C, D at the sizes they are quoted on): libgfship
through the C ABI against the CPU oracle on the same inputs, bit for bit.  The oracle needs tens of
seconds per case at these sizes (0.8 Mcell-steps/s, 40 Mcell-sweeps/s on one core), so every case
runs once and checks everything it can on the way.

  config B  128^3 Dirichlet Poisson solve, 4 V-cycles (single sweeps with the BC kernel around
            them on 8x8 tiles, the coarse end in LDS)
  config C  256^3 triply periodic: one V-cycle on random fields (16x16 tiles, the fused relax
            loops of the 256^3 .. 32^3 levels) and one full Taylor-Green time step after
            simulation_run's start-up (the tiled Godunov kernels on 8x64x64 tiles, fused
            projection updates, CFL)
  config D  10^5 tracers over 24 events on a 64^3 Taylor-Green box, incl. the re-sort by cell after
            16 events: positions (hence cell indices), ids and list order identical
"""
import ctypes as C

import numpy as np
import pytest

import gfship
from flow_cases import PERIODIC, oracle_taylor_green
from oracle import oracle as O
from particle_cases import lcg_positions_fast
from poisson_cases import dirichlet_case, oracle_dirichlet_solve
from test_gpu_timestep import _assert_same_state, _assert_same_un, _device_sim

pytestmark = pytest.mark.gpu
RTOL_SUM = 1e-12      # summed norms are tree-reduced on the device


@pytest.mark.timeout(600)
def test_config_b_128_dirichlet_solve_vs_oracle():
    level, cycles = 7, 4
    n = 1 << level
    rhs, faces, ex = dirichlet_case(3, level)
    gd = gfship.Domain(3, level)
    P, div, res, dia = (gd.variable() for _ in range(4))
    a = np.zeros((n + 2,) * 3)
    a[1:-1, 1:-1, 1:-1] = rhs
    div.upload(a)
    for d, fv in enumerate(faces):
        P.set_bc(d, gfship.BC_DIRICHLET, fv)
    gd.bc(P)
    gd.poisson_coefficients()
    par = gd.params()
    par.tolerance, par.nitermin, par.nitermax = 1e-30, cycles, cycles
    gd.poisson_solve(par, P, div, res, dia, 1.)
    od, oP, ores, opar, _ = oracle_dirichlet_solve(3, level, cycles)
    assert par.niter == opar.niter == cycles
    assert par.residual.infty == opar.residual.infty
    assert par.residual_before.infty == opar.residual_before.infty
    for k in ("first", "second"):
        assert getattr(par.residual, k) == pytest.approx(getattr(opar.residual, k), rel=RTOL_SUM)
    assert np.array_equal(oP.interior(), P.download()[1:-1, 1:-1, 1:-1])
    assert np.array_equal(ores.interior(), res.download()[1:-1, 1:-1, 1:-1])
    # four cycles from P = 0 are on their way to the analytic solution
    assert np.abs(oP.interior() - ex).max() < 2e-2
    gd.destroy()


@pytest.mark.timeout(900)
def test_config_c_256_periodic_vcycle_vs_oracle():
    L = O.lib()
    level, dim = 8, 3
    rng = np.random.default_rng(256)
    od = O.Domain(dim, level, PERIODIC)
    gd = gfship.Domain(dim, level, PERIODIC)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    of, gf = {}, {}
    for nm in ("u", "rhs", "dia", "res"):
        of[nm], gf[nm] = od.field(), gd.variable()
    for nm in ("u", "rhs"):
        a = rng.standard_normal(of[nm].leaf().shape)
        if nm == "rhs":
            a[1:-1, 1:-1, 1:-1] -= a[1:-1, 1:-1, 1:-1].mean()
        of[nm].leaf()[...] = a
        gf[nm].upload(a)
    L.go_bc(of["u"].ptr, of["u"].ptr, level)
    gd.bc(gf["u"])
    L.go_residual(od.ptr, dim, level, of["u"].ptr, of["rhs"].ptr, of["dia"].ptr, of["res"].ptr)
    gd.residual(gf["u"], gf["rhs"], gf["dia"], gf["res"])
    opar, gpar = od.params(), gd.params()
    opar.depth = gpar.depth = level
    L.go_poisson_cycle(od.ptr, C.byref(opar), of["u"].ptr, of["rhs"].ptr, of["dia"].ptr,
                       of["res"].ptr)
    gd.poisson_cycle(gpar, gf["u"], gf["rhs"], gf["dia"], gf["res"])
    gu, gr = gf["u"].download(), gf["res"].download()
    assert np.array_equal(of["u"].interior(), gu[1:-1, 1:-1, 1:-1])
    assert np.array_equal(of["res"].interior(), gr[1:-1, 1:-1, 1:-1])
    # the face ghosts the cycle's last BC application leaves in u
    ou = of["u"].leaf()
    for ax in range(3):
        for s in (0, -1):
            sl = [slice(1, -1)] * 3
            sl[ax] = s
            assert np.array_equal(ou[tuple(sl)], gu[tuple(sl)]), (ax, s)
    on = L.go_norm_residual(od.ptr, 1., of["res"].ptr)
    gn = gd.norm_residual(gf["res"], 1.)
    assert gn.infty == on.infty
    gd.destroy()


@pytest.mark.timeout(1500)
def test_config_c_256_taylor_green_step_vs_oracle():
    level = 8
    osim = oracle_taylor_green(level)
    gd, gs = _device_sim(osim, PERIODIC)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    # config D at its full size: 2e6 tracers on the device (the list bench.py times), and the oracle on
    # a subset of 2e4 of them with the same ids -- tracers are independent of each other, so the
    # subset pins the full-size kernel: one event on the start-up field, one on the field after the step
    npart, stride = 2000000, 100
    pos, ids = lcg_positions_fast(npart)
    gpl = gfship.ParticleList(gs, pos, ids)
    opl = O.Particles(osim, pos[::stride].copy(), ids[::stride].copy())

    def same_tracers(what):
        op, oi = opl.state()
        gp, gi = gpl.download()
        assert len(gi) == gpl.count() and len(np.unique(gi)) == len(gi)
        order = np.argsort(gi, kind="stable")
        at = order[np.searchsorted(gi[order], oi)]
        assert np.array_equal(gi[at], oi), what          # every tracer of the subset is in the device list
        assert np.array_equal(gp[at], op), (what, np.abs(gp[at] - op).max())
        # ... and the device dropped exactly the tracers of the subset that the oracle dropped
        assert np.array_equal(np.isin(ids[::stride], gi), np.isin(ids[::stride], oi)), what

    opl.event()
    gpl.event()
    same_tracers("event 0")
    osim.step()
    gs.step()
    _assert_same_state(osim, gs, "step 0")
    _assert_same_un(osim, gs, "step 0")
    opl.event()
    gpl.event()
    same_tracers("event 1")
    assert gpl.count() > 0.99 * npart
    assert gs.projection_params.niter == osim.projection_params.niter
    assert gs.approx_projection_params.niter == osim.approx_projection_params.niter
    assert gs.projection_params.residual.infty == osim.projection_params.residual.infty
    assert gs.approx_projection_params.residual.infty == \
        osim.approx_projection_params.residual.infty
    assert gs.cfl() == O.lib().go_domain_cfl(osim.ptr)
    gd.destroy()


@pytest.mark.timeout(900)
def test_viscous_taylor_green_128_pipelined_diffusion_sweeps_vs_oracle():
    """implicit viscosity in 3-D at a size where the diffusion multigrid (gfs_diffusion_cycle,
    src/poisson.c:1558-1690) runs its sweeps on the pipelined tile kernels with the diffusion_relax
    cell update: the ring kernel (2 x 2 lines per lane) at 128^3, the six-wave kernel at 64^3 and
    32^3, the LDS loop below -- one start-up and one full time step, bit for bit"""
    level = 7
    osim = oracle_taylor_green(level)
    for c in range(3):
        osim.set_viscosity(c, 2e-3)
    gd, gs = _device_sim(osim, PERIODIC)
    for c in range(3):
        gs.set_viscosity(c, 2e-3)
    osim.start()
    gs.start()
    _assert_same_state(osim, gs, "start")
    osim.step()
    gs.step()
    _assert_same_state(osim, gs, "step 0")
    for c in range(3):
        a, b = osim.diffusion_params(c), gs.diffusion_params(c)
        assert a.niter == b.niter and a.niter >= 1
        assert a.residual.infty == b.residual.infty
        assert a.residual_before.infty == b.residual_before.infty
    gd.destroy()


@pytest.mark.timeout(900)
def test_config_d_1e5_tracers_24_events_with_resort_vs_oracle():
    level, npart, nev = 6, 100000, 24
    osim = oracle_taylor_green(level)
    # a mean flow so that many tracers cross cells, box sides and the periodic wrap
    osim.u[0].interior()[...] += 0.9
    osim.u[2].interior()[...] -= 0.4
    gd, gs = _device_sim(osim, PERIODIC)
    pos, ids = lcg_positions_fast(npart)
    osim.start()
    gs.start()
    opl = O.Particles(osim, pos, ids)
    gpl = gfship.ParticleList(gs, pos, ids)      # default: slots re-sorted by cell every 16 events
    n = 1 << level
    for k in range(nev):
        opl.event()
        gpl.event()
        if k % 4 == 3 or k in (15, 16, 17):
            op, oi = opl.state()
            gp, gi = gpl.download()
            assert np.array_equal(oi, gi), k
            assert np.array_equal(op, gp), (k, np.abs(op - gp).max())
            # cell indices (level, i, j, k) follow from the positions with the reference's locate
            # rule (strict > against the cell centres, src/ftt.c:1563): identical by construction,
            # checked explicitly
            ci = np.floor((gp + 0.5) * n).astype(np.int64)
            co = np.floor((op + 0.5) * n).astype(np.int64)
            assert np.array_equal(ci, co)
        osim.step()
        gs.step()
    # (a particle that leaves through an edge or a corner of the box is not wrapped but dropped by
    # gfs_particle_bc, modules/particulatecommon.c:3151-3214: the lists shrink together)
    assert gpl.count() == opl.count() and gpl.count() > 0.9 * npart
    # the wrap moved some of them (periodic_bc_particle, modules/particulatecommon.c:3189-3214)
    gp, gi = gpl.download()
    assert np.abs(gp - pos[gi.astype(np.int64) - 1]).max() > 0.5
    gd.destroy()
