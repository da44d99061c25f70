"""Energy spectra (modules/fft.c:1340-1474): properties of the numpy restatement
oracle/go_spectra.py.  The reference holds no golden data for its FFT outputs and FFTW is absent:
parity UNPINNED; these are the analytic pins."""
import numpy as np

from oracle.go_spectra import energy_spectra


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
