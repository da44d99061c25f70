"""Flow set-ups shared by oracle tests and GPU parity tests."""
import numpy as np

from oracle import oracle as O

PERIODIC = [O.SIDE_PERIODIC] * 6


def reynolds_init(x, y):
    # test/reynolds/reynolds.gfs:67-72
    return -np.cos(2. * np.pi * x) * np.sin(2. * np.pi * y), np.sin(2. * np.pi * x) * np.cos(2. * np.pi * y)


def oracle_reynolds(level):
    s = O.Sim(2, level, PERIODIC)
    x, y = s.dom.centres()
    u, v = reynolds_init(x, y)
    s.u[0].interior()[...] = u
    s.u[1].interior()[...] = v
    s.approx_projection_params.tolerance = 1e-6
    s.projection_params.tolerance = 1e-6
    s.set_time(end=2.)
    return s


def periodic_init(x, y):
    # test/periodic/periodic.gfs:53-56
    return (1. - 2. * np.cos(2. * np.pi * x) * np.sin(2. * np.pi * y),
            1. + 2. * np.sin(2. * np.pi * x) * np.cos(2. * np.pi * y))


def oracle_periodic(level):
    s = O.Sim(2, level, PERIODIC)
    x, y = s.dom.centres()
    u, v = periodic_init(x, y)
    s.u[0].interior()[...] = u
    s.u[1].interior()[...] = v
    s.advection_params.cfl = 0.75
    s.approx_projection_params.tolerance = 1e-6
    s.projection_params.tolerance = 1e-6
    s.set_time(end=0.5)
    return s


def taylor_green_3d(x, y, z):
    # SURVEY.md 8d config C
    u = np.sin(2. * np.pi * x) * np.cos(2. * np.pi * y) * np.cos(2. * np.pi * z)
    v = -np.cos(2. * np.pi * x) * np.sin(2. * np.pi * y) * np.cos(2. * np.pi * z)
    w = np.zeros(np.broadcast(x, y, z).shape)
    return u + 0. * w, v + 0. * w, w


def oracle_taylor_green(level):
    s = O.Sim(3, level, PERIODIC)
    x, y, z = s.dom.centres()
    for c, a in enumerate(taylor_green_3d(x, y, z)):
        s.u[c].interior()[...] = a
    return s


def oracle_lid(level=6, nu=1e-3):
    # test/lid/lid.gfs: unit box, Dirichlet walls, lid U = 1 on top, SourceDiffusion U/V nu
    s = O.Sim(2, level, [O.SIDE_BOUNDARY] * 6)
    n = 1 << level
    for c in range(2):
        for d in range(4):
            val = np.full(n, 1. if (c == 0 and d == 2) else 0.)
            s.u[c].set_bc(d, O.BC_DIRICHLET, val)
        s.set_viscosity(c, nu)
    s.set_time(end=300.)
    return s


def run_until_steady(s, var, every=10, tol=1e-4, max_steps=100000):
    """simulation_run with GfsEventStop { istep = every } var tol (src/event.c:1797-1835): every
    `every` steps compare var with its previous copy; when max|diff| <= tol the event sets
    time.end = time.t, the loop body still completes (src/simulation.c:476-548) and the loop
    condition ends the run."""
    s.start()
    old = None
    hist = []
    end = s.end
    while s.t < end and s.i < max_steps:
        if s.i % every == 0:
            cur = var.interior().copy()
            if old is not None:
                du = np.abs(cur - old).max()
                hist.append((s.i, s.t, du))
                if du <= tol:
                    end = s.t
            old = cur
        s.step()
    return hist
