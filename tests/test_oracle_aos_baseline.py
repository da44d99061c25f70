"""The AoS pointer-tree variant of the CPU baseline (oracle/go_aos.c: FttCell / FttOct records,
ftt_cell_neighbor lookups, recursive traversal with a callback per cell) reproduces the flat-array
oracle's relax sweeps bit for bit: same arithmetic, same order, different memory layout."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O


def _sigs(L):
    vp, pd, i = C.c_void_p, C.POINTER(C.c_double), C.c_int
    L.go_aos_new.restype, L.go_aos_new.argtypes = vp, [i]
    L.go_aos_destroy.restype, L.go_aos_destroy.argtypes = None, [vp]
    L.go_aos_load.restype, L.go_aos_load.argtypes = None, [vp, vp, pd, pd, pd]
    L.go_aos_store.restype, L.go_aos_store.argtypes = None, [vp, vp, pd]
    L.go_aos_relax.restype, L.go_aos_relax.argtypes = None, [vp, i]
    L.go_aos_bytes_per_cell.restype = C.c_size_t


@pytest.mark.parametrize("level,with_dia", [(3, False), (4, True), (5, False)])
def test_aos_tree_relax_equals_flat_oracle(level, with_dia):
    L = O.lib()
    _sigs(L)
    dom = O.Domain(3, level, [O.SIDE_PERIODIC] * 6)
    L.go_poisson_coefficients(dom.ptr)
    rng = np.random.default_rng(level)
    u, rhs, dia = dom.field(), dom.field(), dom.field()
    u.interior()[...] = rng.standard_normal(u.interior().shape)
    rhs.interior()[...] = rng.standard_normal(u.interior().shape)
    if with_dia:
        dia.interior()[...] = np.abs(rng.standard_normal(u.interior().shape))
    pd = C.POINTER(C.c_double)
    lev = lambda f: C.cast(L.go_field_level(f.ptr, level), pd)
    tree = L.go_aos_new(level)
    L.go_aos_load(tree, dom.ptr, lev(u), lev(rhs), lev(dia))
    nsweeps = 3
    L.go_aos_relax(tree, nsweeps)
    for _ in range(nsweeps):
        L.go_homogeneous_bc(u.ptr, u.ptr, level)
        L.go_relax(dom.ptr, 3, level, 1., u.ptr, rhs.ptr, dia.ptr)
    want = u.interior().copy()
    L.go_aos_store(tree, dom.ptr, lev(u))
    assert np.array_equal(u.interior(), want)
    assert L.go_aos_bytes_per_cell() > 200          # a reference-like record, not 8 bytes per variable
    L.go_aos_destroy(tree)
