"""oracle/go_spectra.py -- CPU restatement of GfsOutputEnergySpectra (modules/fft.c:1340-1474) for a
uniform periodic box.  TEST INFRASTRUCTURE ONLY (see gfs_oracle.h).

The FFT arithmetic of the reference lives in FFTW3 (>= 3, with fftw3_mpi; un-vendored, absent here:
`modules/fft.h:26`, configure.ac:361-367) and the reference holds no test or golden data for its
spectra: PARITY UNPINNED.  What is restated is the published algorithm of the call sites:
  fill_cartesian_matrix (:966-1001)  v = (u - <u>)/ntot on the np^dim grid of the finest level
                                     (substract_average :897-908, get_data :399-417)
  get_fftw_plan (:1087-1098)          real-to-complex DFT over all dimensions (last one halved)
  output_energy_spectra_event (:1360-1474)
      3-D  Ek[knx^2 + kny^2 + k^2] += w |F[i][j][k]|^2, w = 1/2 for k = 0, 1 for k = 1 .. np/2,
           knx = i if i < np/2 + 1 else np - i (same for kny)
      2-D  Ek[knx^2] += 1/2 |F[i][0]|^2, then Ek[knx^2 + j^2] += |F[i][j]|^2 for j = 0 .. np/2
           (j = 0 is counted again: as in the reference)
      nk = (dim + 1) (np/2 + 1)^2 bins, Etot = sum of all bins
  write_energy_spectra (:1340-1348)   lines "deltak*sqrt(i) Ek[i]" for i = 1 .. nk - 1, with
      deltak = 2 pi/(x1 - x0) and x0, x1 the first / last CELL CENTRES along x ((np - 1) dx apart).
The order of the dimensions of the transform (the reference sorts them by size, :800-820, all equal
for a cube) is fixed here to x, y, z with z halved; the components are summed in the order U, V, W.
Uses numpy's pocketfft; sums in the order of the loops above."""
import numpy as np


def energy_spectra(comps):
    """comps: interior arrays of the velocity components, indexed [k][j][i] (3-D) or [j][i] (2-D)
    as oracle.Field.interior() returns them.  Returns (k, Ek, Etot): the printed lines and the
    total energy."""
    dim = comps[0].ndim
    n = comps[0].shape[0]
    nk = (dim + 1) * (n // 2 + 1) ** 2
    Ek = np.zeros(nk)
    kn = np.where(np.arange(n) < n // 2 + 1, np.arange(n), n - np.arange(n))
    for u in comps:
        a = np.transpose(u)                       # [ix][iy]([iz])
        a = (a - a.mean()) / float(n ** dim)
        F = np.fft.rfftn(a)
        P = F.real ** 2 + F.imag ** 2
        if dim == 3:
            kz = np.arange(n // 2 + 1)
            bins = (kn[:, None, None] ** 2 + kn[None, :, None] ** 2 + kz[None, None, :] ** 2)
            w = np.where(kz == 0, 0.5, 1.)[None, None, :]
            np.add.at(Ek, bins.ravel(), (w * P).ravel())
        else:
            ky = np.arange(n // 2 + 1)
            np.add.at(Ek, kn ** 2, 0.5 * P[:, 0])
            bins = kn[:, None] ** 2 + ky[None, :] ** 2
            np.add.at(Ek, bins.ravel(), P.ravel())
    Etot = float(Ek.sum())
    dx = 1. / n
    deltak = 2. * np.pi / ((n - 1) * dx)
    i = np.arange(1, nk)
    return deltak * np.sqrt(i.astype(float)), Ek[1:].copy(), Etot
