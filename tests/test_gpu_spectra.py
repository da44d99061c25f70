"""GfsOutputEnergySpectra on the device (hipFFT + the reference's binning, spectra.hip) against the
numpy restatement of the reference's call sites (oracle/go_spectra.py).  Tolerance 1e-12 of the
total energy per bin: two FFT libraries and atomic sums, same definition."""
import numpy as np
import pytest

import gfship
from oracle.go_spectra import energy_spectra

pytestmark = pytest.mark.gpu


def _with_ghosts(a):
    return np.pad(a, 1, mode="wrap")


@pytest.mark.parametrize("dim,level", [(3, 5), (3, 6), (2, 6), (2, 8)])
def test_energy_spectra_match_the_restatement(dim, level):
    n = 1 << level
    rng = np.random.default_rng(100 * dim + level)
    gd = gfship.Domain(dim, level, [gfship.SIDE_PERIODIC] * 6)
    comps, fields = [], []
    x = (np.arange(n) + 0.5) / n - 0.5
    grids = np.meshgrid(*([x] * dim), indexing="ij")[::-1]
    for c in range(dim):
        a = 0.3 + rng.standard_normal((n,) * dim) * 0.1 + np.sin(2 * np.pi * (c + 1) * grids[c])
        f = gd.variable()
        f.upload(_with_ghosts(a))
        comps.append(a)
        fields.append(f)
    k, Ek, Etot = gd.energy_spectra(fields)
    k0, Ek0, Etot0 = energy_spectra(comps)
    assert np.array_equal(k, k0)
    assert np.isclose(Etot, Etot0, rtol=1e-12)
    assert np.abs(Ek - Ek0).max() <= 1e-12 * Etot0
    # the imposed modes dominate their bins
    for c in range(dim):
        assert Ek[(c + 1) ** 2 - 1] > 0.2


def test_energy_spectra_of_the_taylor_green_field_256():
    """full size: one mode per component, |k|^2 = 3, total = kinetic energy per unit volume"""
    level, n = 8, 256
    x = (np.arange(n) + 0.5) / n - 0.5
    Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
    u = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z)
    v = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.cos(2 * np.pi * Z)
    gd = gfship.Domain(3, level, [gfship.SIDE_PERIODIC] * 6)
    fs = []
    for a in (u, v, 0 * u):
        f = gd.variable()
        f.upload(_with_ghosts(a))
        fs.append(f)
    k, Ek, Etot = gd.energy_spectra(fs)
    assert np.isclose(Etot, 0.5 * (np.mean(u ** 2) + np.mean(v ** 2)), rtol=1e-12)
    assert np.isclose(Ek[3 - 1], Etot, rtol=1e-12)
    assert np.abs(np.delete(Ek, 2)).max() < 1e-25
