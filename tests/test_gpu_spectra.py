"""GfsOutputEnergySpectra on the device (hipFFT + the reference's binning, spectra.hip) against the
numpy restatement of the reference's call sites (oracle/go_spectra.py).  Tolerance 1e-12 of the
total energy per bin: two FFT libraries and atomic sums, same definition."""
import numpy as np
import pytest

import gfship
from oracle.go_spectra import energy_spectra, output_spectra, turbulent_viscosity

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


POPE = dict(x0=0., y0=0., z0=0., L=1., E=1.5, alpha=1.5, epsilon=1., c1=6.78, c2=5.2, c3=0.4,
            ReL=1000., kmax=1e30, seed=7., level=5)


@pytest.mark.parametrize("par_update,level", [({}, 5), ({"ReL": 0., "kmax": 60.}, 5), ({"level": 4}, 5),
                                              ({"seed": 123.}, 6)])
def test_init_spectra_matches_the_restatement(par_update, level):
    """GfsInitSpectra: the device field (host loops in the reference's order + hipFFT + interpolation
    kernel) against the numpy restatement, relative to the rms velocity"""
    from oracle.go_spectra import init_spectra
    par = dict(POPE)
    par["level"] = level
    par.update(par_update)
    n = 1 << level
    gd = gfship.Domain(3, level, [gfship.SIDE_PERIODIC] * 6)
    fs = [gd.variable(c) for c in range(3)]
    gd.init_spectra(par, fs)
    want = init_spectra(par, n)
    rms = np.sqrt(np.mean(sum(a ** 2 for a in want)))
    assert rms > 0.1
    for c in range(3):
        got = fs[c].download()[1:-1, 1:-1, 1:-1]
        assert np.abs(got - want[c]).max() <= 1e-10 * rms, c


def test_init_spectra_field_properties_128():
    """the synthetic field at 128^3: shell energies follow the target model spectrum where the
    trilinear interpolation onto the cell centres does not damp them (low wavenumbers), the total
    energy is close to E, and the field is nearly solenoidal at the resolved scales"""
    level, n = 7, 128
    par = dict(POPE)
    par["level"] = level
    gd = gfship.Domain(3, level, [gfship.SIDE_PERIODIC] * 6)
    fs = [gd.variable(c) for c in range(3)]
    gd.init_spectra(par, fs)
    k, Ek, Etot = gd.energy_spectra(fs)
    assert 0.8 * par["E"] < Etot < 1.05 * par["E"]
    # target shell energies at the first shells: Ei = cscale2^2 E(k_i) with sum Ei = E
    deltak = 2 * np.pi / par["L"]
    i = np.arange(1, 3 * (n // 2 + 1) ** 2)
    kw = deltak * np.sqrt(i.astype(float))
    Lint = par["E"] ** 1.5 / par["epsilon"]
    fl = (Lint * kw / np.sqrt((Lint * kw) ** 2 + par["c1"])) ** (11. / 3.)
    feta = np.exp(-par["c2"] * (((Lint * kw * par["ReL"] ** (-0.75)) ** 4 + par["c3"] ** 4) ** 0.25 - par["c3"]))
    Ei = par["alpha"] * par["epsilon"] ** (2. / 3.) * kw ** (-5. / 3.) * fl * feta
    # shells that exist on the lattice (sums of three squares)
    from oracle.go_spectra import energy_spectra, output_spectra, turbulent_viscosity  # noqa: F401  (restatement used by the other tests)
    exists = Ek[:len(i)] > 0
    Ei = np.where(exists[:len(Ei)], Ei, 0.)
    Ei *= par["E"] / Ei.sum()
    low = slice(0, 12)
    # (the grid of the c2r transform spans L with np - 1 intervals while the cells span it with np:
    # the modes are slightly stretched and leak between the integer shells; 25 % is what is left)
    assert np.allclose(Ek[low][exists[low]], Ei[low][exists[low]], rtol=0.25)
    assert np.all(np.diff(Ek[low][exists[low]]) < 0.)


@pytest.mark.parametrize("level", [4, 6])
def test_output_spectra_matches_the_restatement(level):
    """GfsOutputSpectra (modules/fft.c:1101-1160): the r2c transform of one variable of the whole box"""
    n = 1 << level
    rng = np.random.default_rng(level)
    x = (np.arange(n) + 0.5) / n - 0.5
    Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
    a = 1.7 + 0.1 * rng.standard_normal((n, n, n)) + np.cos(2 * np.pi * 2 * X) + 0.5 * np.sin(2 * np.pi * 3 * Z)
    gd = gfship.Domain(3, level, [gfship.SIDE_PERIODIC] * 6)
    f = gd.variable()
    f.upload(_with_ghosts(a))
    F, ks = gd.output_spectra(f)
    F0, ks0 = output_spectra(a)
    assert ks == ks0
    assert F.shape == F0.shape
    assert np.abs(F - F0).max() <= 1e-13
    assert np.isclose(abs(F[2, 0, 0]), 0.5, atol=1e-2) and np.isclose(abs(F[0, 0, 3]), 0.25, atol=1e-2)
    gd.destroy()


@pytest.mark.parametrize("dim,model", [(3, 1), (2, 1), (3, 0), (2, 0)])
def test_turbulent_viscosity_matches_the_restatement(dim, model):
    """GfsVariableTurbulentViscosity: Smagorinsky (what a simulation file gets) bit for bit up to the
    square root; the sigma model through the device's acos / cos / pow within 1e-10 of its scale"""
    level = 5
    n = 1 << level
    rng = np.random.default_rng(7 + dim)
    gd = gfship.Domain(dim, level, [gfship.SIDE_PERIODIC] * 6)
    u, fields = [], []
    for c in range(dim):
        a = _with_ghosts(rng.standard_normal((n,) * dim))
        f = gd.variable()
        f.upload(a)
        u.append(a)
        fields.append(f)
    out = gd.variable()
    gd.turbulent_viscosity(fields, 0.2, out, model=model)
    got = out.download()[(slice(1, -1),) * dim]
    want = turbulent_viscosity(u, 0.2, model=model)
    if model == 1:
        assert np.abs(got - want).max() <= 4e-16 * np.abs(want).max()
    else:
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
        assert (got > 0).any()
    gd.destroy()


@pytest.mark.parametrize("normal,pos", [(2, 0.1), (0, -0.3), (1, 0.26)])
def test_output_spectra_of_a_plane_matches_the_restatement(normal, pos):
    """GfsOutputSpectra with a flat box (a plane of the 3-D box): the 2-D r2c DFT of the cell values on the
    plane against the numpy restatement of modules/fft.c:822-883,1101-1160, and a single in-plane mode"""
    from oracle import go_spectra as GS
    level = 5
    n = 1 << level
    gd = gfship.Domain(3, level, [gfship.SIDE_PERIODIC] * 6)
    v = gd.variable()
    rng = np.random.default_rng(12)
    a = np.zeros((n + 2,) * 3)
    c = -0.5 + (np.arange(1, n + 1) - 0.5) / n
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    a[1:-1, 1:-1, 1:-1] = np.sin(2. * np.pi * 3. * x) * np.cos(2. * np.pi * 2. * y) * (1. + z) + \
        0.1 * rng.standard_normal((n,) * 3)
    v.upload(a)
    F, ks = gd.output_spectra_plane(v, normal, pos)
    Fo, kso = GS.output_spectra_plane(a[1:-1, 1:-1, 1:-1], normal, pos)
    assert ks == kso
    assert np.abs(F - Fo).max() <= 1e-13 * max(1., np.abs(Fo).max())
    assert abs(F[0, 0]) <= 1e-15                   # the mean of the plane has been removed
    gd.destroy()
