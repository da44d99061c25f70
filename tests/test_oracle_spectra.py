"""Energy spectra (modules/fft.c:1340-1474): properties of the numpy restatement
oracle/go_spectra.py.  The reference holds no golden data for its FFT outputs and FFTW is absent:
parity UNPINNED; these are the analytic pins."""
import numpy as np

from oracle.go_spectra import all_level_mean, energy_spectra, output_spectra, turbulent_viscosity


def _grid(n, dim):
    x = (np.arange(n) + 0.5) / n - 0.5
    return np.meshgrid(*([x] * dim), indexing="ij")[::-1]      # X fastest: arrays indexed [k][j][i]


def test_single_modes_land_in_their_bins_3d():
    n = 32
    X, Y, Z = _grid(n, 3)
    u = np.sin(2 * np.pi * 3 * X)
    v = 0.5 * np.cos(2 * np.pi * 2 * Y) * np.sin(2 * np.pi * 1 * Z)
    k, Ek, Etot = energy_spectra([u, v, 0 * u])
    nz = np.nonzero(Ek > 1e-20)[0] + 1
    assert list(nz) == [5, 9]                                  # 2^2 + 1^2 and 3^2
    assert np.isclose(Ek[9 - 1], 0.5 * np.mean(u ** 2), rtol=1e-13)
    assert np.isclose(Ek[5 - 1], 0.5 * np.mean(v ** 2), rtol=1e-13)
    # Parseval: the total is the kinetic energy per unit volume (mean removed)
    assert np.isclose(Etot, 0.5 * (np.mean(u ** 2) + np.mean(v ** 2)), rtol=1e-13)
    # abscissae: deltak sqrt(i), deltak = 2 pi / distance between the first and last cell centres
    assert np.isclose(k[0], 2 * np.pi / ((n - 1) / n)) and np.isclose(k[8], 3 * k[0])
    assert len(Ek) == 4 * (n // 2 + 1) ** 2 - 1


def test_mean_flow_is_removed_and_band_limited_parseval():
    n = 16
    rng = np.random.default_rng(3)
    X, Y, Z = _grid(n, 3)
    u = 2.5 + sum(rng.standard_normal() * np.sin(2 * np.pi * (a * X + b * Y + c * Z) + rng.random())
                  for a, b, c in [(1, 0, 0), (1, 2, 0), (0, 3, 1), (2, 2, 2), (5, 1, 3)])
    _, _, Etot = energy_spectra([u])
    assert np.isclose(Etot, 0.5 * np.var(u), rtol=1e-12)       # no Nyquist content: exact Parseval


def test_2d_overweights_the_ky_zero_line_like_the_reference():
    n = 32
    X, Y = _grid(n, 2)
    # ky = 0: both conjugate modes are in the half-complex output and each gets 1/2 + 1 (:1424-1432)
    ux = np.sin(2 * np.pi * 3 * X)
    uy = np.sin(2 * np.pi * 3 * Y)                              # ky = 3: weight 1
    _, Ex, _ = energy_spectra([ux])
    _, Ey, _ = energy_spectra([uy])
    assert np.isclose(Ex[9 - 1], 3.0 * 0.5 * np.mean(ux ** 2), rtol=1e-13)
    assert np.isclose(Ey[9 - 1], 1.0 * 0.5 * np.mean(uy ** 2), rtol=1e-13)


def test_output_spectra_single_mode_and_parseval():
    """GfsOutputSpectra: a cosine along x and a sine along z land at their wave vectors with
    amplitude 1/2 each (r2c of v/ntot), the mean is removed, Parseval holds for the half spectrum"""
    n = 16
    x = (np.arange(n) + 0.5) / n - 0.5
    Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
    v = 3. + np.cos(2 * np.pi * 2 * X) + 0.5 * np.sin(2 * np.pi * 3 * Z)
    F, ks = output_spectra(v)
    assert F.shape == (n, n, n // 2 + 1)
    assert np.isclose(ks, 2 * np.pi / (1. - 1. / n))
    assert abs(F[0, 0, 0]) < 1e-15
    assert np.isclose(abs(F[2, 0, 0]), 0.5) and np.isclose(abs(F[n - 2, 0, 0]), 0.5)
    assert np.isclose(abs(F[0, 0, 3]), 0.25)
    w = np.full(n // 2 + 1, 2.)
    w[0] = w[-1] = 1.
    assert np.isclose((w * (F.real ** 2 + F.imag ** 2)).sum(), ((v - v.mean()) ** 2).mean())


def test_all_level_mean_weights_every_level_equally():
    rng = np.random.default_rng(2)
    v = rng.standard_normal((8, 8, 8))
    assert np.isclose(all_level_mean(v), v.mean(), rtol=1e-14, atol=1e-16)


def test_smagorinsky_of_a_pure_shear_and_sigma_of_solid_rotation():
    """u = (S y, 0, 0): |S| = sqrt (2 S_ij S_ij) = S, nu = (Cs h)^2 S; the sigma model vanishes for
    pure shear and for solid rotation (its defining property), Smagorinsky does not for shear"""
    n, Cs, S = 8, 0.17, 2.5
    x = (np.arange(-1, n + 1) + 0.5) / n - 0.5
    Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
    u = [S * Y, 0. * X, 0. * X]
    nu = turbulent_viscosity(u, Cs)
    assert np.allclose(nu, (Cs / n) ** 2 * S, rtol=1e-13)
    assert np.abs(turbulent_viscosity(u, Cs, model=0)).max() < 1e-12
    rot = [-Y, X, 0. * X]
    assert np.abs(turbulent_viscosity(rot, Cs, model=0)).max() < 1e-12
    assert np.abs(turbulent_viscosity(rot, Cs)).max() < 1e-15       # S_ij = 0
