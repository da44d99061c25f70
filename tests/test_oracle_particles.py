"""The reference holds no test, example or golden file for its particle module:
parity of the tracer path is UNPINNED by the reference.  The oracle's restatement of
locate / interpolate / RK2 / periodic wrap is checked here on analytic properties."""
import numpy as np
import pytest

from flow_cases import PERIODIC
from oracle import oracle as O
from particle_cases import lcg_positions


def _uniform_flow(dim, level, vel, side=PERIODIC):
    s = O.Sim(dim, level, side)
    for c in range(dim):
        s.u[c].leaf()[...] = vel[c]        # ghosts included: a uniform field
    return s


def test_locate_matches_cell_containing_point_and_tie_rule():
    s = _uniform_flow(3, 4, (0., 0., 0.))
    pl = O.Particles(s, np.zeros((1, 3)), [1])
    n, h = 16, 1. / 16
    rng = np.random.default_rng(0)
    for p in rng.uniform(-0.5, 0.5, (200, 3)):
        ijk = pl.locate(p)
        for c in range(3):
            lo = -0.5 + (ijk[c] - 1) * h
            assert lo <= p[c] <= lo + h
    # strict '>' comparisons of ftt_cell_locate (ftt.c:1563): a point on an interior cell face
    # belongs to the lower cell, the box bounds are inclusive (:1547-1553)
    assert pl.locate([0., 0., 0.]) == (8, 8, 8)
    assert pl.locate([h, 0., 0.]) == (9, 8, 8)
    assert pl.locate([0.5, 0.5, 0.5]) == (16, 16, 16)
    assert pl.locate([-0.5, -0.5, -0.5]) == (1, 1, 1)
    assert pl.locate([0.5 + 1e-12, 0., 0.]) is None


@pytest.mark.parametrize("dim", [2, 3])
def test_uniform_flow_translates_exactly(dim):
    vel = (0.3, -0.2, 0.1)[:dim]
    s = _uniform_flow(dim, 4, vel)
    s.advection_params.dt = 0.05
    pos, ids = lcg_positions(300, dim)
    pl = O.Particles(s, pos, ids)
    nsteps = 10
    for _ in range(nsteps):
        pl.event()
    x1, ids1 = pl.state()
    disp = np.array(vel + (0.,) * (3 - dim)) * 0.05 * nsteps
    x0s = pos[ids1 - 1]
    # particles whose whole path stays inside the box move by exactly u*dt per step
    inside = np.all(np.abs(x0s + disp)[:, :dim] < 0.49, axis=1) & np.all(np.abs(x0s)[:, :dim] < 0.49, axis=1)
    assert inside.sum() > 100
    assert np.abs(x1[inside] - x0s[inside] - disp).max() < 1e-14


def test_periodic_wrap_and_midpoint_rule():
    """gfs_domain_advect_point leaves a point unchanged when its RK2 midpoint is outside the
    domain (domain.c:2783-2785); a point whose end position is outside is wrapped by
    periodic_bc_particle with the 1e-8 box-size nudge (particulatecommon.c:3204-3210) and gets
    pos_old = pos."""
    u, dt = 0.4, 0.1
    s = _uniform_flow(3, 4, (u, 0., 0.))
    s.advection_params.dt = dt
    x_cross = 0.5 - 0.6 * dt * u      # midpoint inside, end point outside: wraps
    x_stuck = 0.5 - 0.3 * dt * u      # midpoint outside: does not move at all
    pl = O.Particles(s, [[x_cross, 0.1, 0.2], [x_stuck, 0.1, 0.2]], [1, 2])
    pl.event()
    x1, ids = pl.state()
    assert list(ids) == [1, 2]
    expect = -0.5 + ((x_cross + dt * u) - 0.5) * 1. + 1. * (1. / 1.e8)
    assert x1[0, 0] == expect and x1[0, 1] == 0.1 and x1[0, 2] == 0.2
    assert x1[1, 0] == x_stuck


def test_interpolation_reproduces_linear_fields_in_the_interior():
    """inverse-distance corner weights + the 8-corner polynomial are exact for linear fields
    away from the box sides"""
    s = O.Sim(3, 4, PERIODIC)
    n = 16
    c = -0.5 + (np.arange(0, n + 2) - 0.5) / n
    X, Y, Z = c[None, None, :], c[None, :, None], c[:, None, None]
    s.u[0].leaf()[...] = 1. + 2. * X - 3. * Y + 0.5 * Z
    s.u[1].leaf()[...] = 0. * X + 0. * Y + 0. * Z
    s.u[2].leaf()[...] = 0. * X + 0. * Y + 0. * Z
    s.advection_params.dt = 1e-3
    p = np.array([[0.113, -0.207, 0.051], [-0.3, 0.26, 0.33]])
    pl = O.Particles(s, p, [1, 2])
    pl.event()
    x1, _ = pl.state()
    # RK2 midpoint with u = 1 + 2x - 3y + z/2 (v = w = 0): x1 = x0 + dt*u(x0 + dt*u(x0)/2, y, z)
    u0 = 1. + 2. * p[:, 0] - 3. * p[:, 1] + 0.5 * p[:, 2]
    um = 1. + 2. * (p[:, 0] + 1e-3 * u0 / 2.) - 3. * p[:, 1] + 0.5 * p[:, 2]
    assert np.allclose(x1[:, 0], p[:, 0] + 1e-3 * um, rtol=0, atol=1e-15)
    assert np.array_equal(x1[:, 1:], p[:, 1:])


def test_closed_box_drops_particles_that_leave():
    side = [O.SIDE_BOUNDARY] * 6
    s = _uniform_flow(3, 3, (1., 0., 0.), side)
    s.advection_params.dt = 0.1
    pl = O.Particles(s, [[0.45, 0., 0.], [0., 0., 0.]], [1, 2])
    pl.event()
    _, ids = pl.state()
    assert list(ids) == [2]
