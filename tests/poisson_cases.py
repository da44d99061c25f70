"""Problem set-ups shared by the oracle tests and the GPU parity tests."""
import ctypes as C

import numpy as np

from oracle import oracle as O


def exact2d(X, Y):
    # s = sin(3 pi x) sin(3 pi y): OutputErrorNorm / BcDirichlet of test/poisson/poisson.gfs:84-94
    return np.sin(np.pi * 3. * X) * np.sin(np.pi * 3. * Y)


def exact3d(X, Y, Z):
    return np.sin(np.pi * 3. * X) * np.sin(np.pi * 3. * Y) * np.sin(np.pi * 3. * Z)


def dirichlet_case(dim, level):
    """test/poisson/poisson.gfs (2-D) and its 3-D analogue:
    returns rhs = Div*h^2 on the interior and the six/four face-value arrays."""
    n = 1 << level
    h = 1. / n
    c = -0.5 + (np.arange(1, n + 1) - 0.5) / n
    if dim == 2:
        x, y = c[None, :], c[:, None]
        k = l = 3
        div = -np.pi * np.pi * (k * k + l * l) * np.sin(np.pi * k * x) * np.sin(np.pi * l * y)
        faces = [exact2d(0.5, c), exact2d(-0.5, c), exact2d(c, 0.5), exact2d(c, -0.5)]
        ex = exact2d(x, y)
    else:
        x, y, z = c[None, None, :], c[None, :, None], c[:, None, None]
        div = -27. * np.pi * np.pi * exact3d(x, y, z)
        a, b = c[None, :], c[:, None]  # first tangential axis fastest
        faces = [exact3d(0.5, a, b), exact3d(-0.5, a, b),
                 exact3d(a, 0.5, b), exact3d(a, -0.5, b),
                 exact3d(a, b, 0.5), exact3d(a, b, -0.5)]
        ex = exact3d(x, y, z)
    return div * (h * h * 1.), faces, ex


def oracle_dirichlet_solve(dim, level, cycles, tolerance=1e-30):
    """GfsPoisson run (simulation.c:2213-2285) on the oracle."""
    L = O.lib()
    dom = O.Domain(dim, level)
    P, div, res, dia = dom.field(), dom.field(), dom.field(), dom.field()
    rhs, faces, ex = dirichlet_case(dim, level)
    div.interior()[...] = rhs
    for d, f in enumerate(faces):
        P.set_bc(d, O.BC_DIRICHLET, f)
    L.go_bc(P.ptr, P.ptr, level)
    L.go_poisson_coefficients(dom.ptr)
    par = dom.params()
    par.tolerance = tolerance
    par.nitermin = cycles
    par.nitermax = cycles
    L.go_poisson_solve(dom.ptr, C.byref(par), P.ptr, div.ptr, res.ptr, dia.ptr, 1.)
    return dom, P, res, par, ex


def error_norm_unbiased(dom, P, ex):
    """OutputErrorNorm { v = P } { s = exact unbiased = 1 } (output.c:2940-3030)."""
    L = O.lib()
    e = dom.field()
    e.interior()[...] = P.interior() - ex
    nm = L.go_norm_variable(dom.ptr, e.ptr)
    e.interior()[...] -= nm.bias
    return L.go_norm_variable(dom.ptr, e.ptr)
