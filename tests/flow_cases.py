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
