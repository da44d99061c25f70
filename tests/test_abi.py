"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/gfship.h
declares, and fails loudly (no fallback) when there is no device."""
import os
import re

import pytest

import gfship
from conftest import ROOT, has_gpu


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gfship.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfship_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = gfship.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libgfship.so does not export %s" % n
    # and the python mirror binds every one of them
    assert set(names) == set(gfship.SIGNATURES), set(names) ^ set(gfship.SIGNATURES)


def test_version():
    assert gfship.lib().gfship_version() == 1


@pytest.mark.skipif(has_gpu(), reason="checks the no-device error path")
def test_no_device_is_an_error_not_a_fallback():
    with pytest.raises(gfship.GfshipError, match="no HIP device"):
        gfship.Domain(3, 3)
    with pytest.raises(gfship.GfshipError, match="no HIP device"):
        gfship.Tree(lambda x, y: 3)           # the refined-tree path has no host fallback either


def test_multilevel_params_defaults_match_reference():
    # gfs_multilevel_params_init, src/poisson.c:70-89
    import ctypes as C
    p = gfship.MultilevelParams()
    gfship.lib().gfship_multilevel_params_init(C.byref(p), 3)
    assert (p.tolerance, p.nrelax, p.erelax, p.minlevel, p.nitermax, p.nitermin) == \
        (1e-3, 4, 1, 0, 100, 1)
    assert (p.dimension, p.weighted, p.beta, p.omega, p.function) == (3, 0, 1., 1., 0)
