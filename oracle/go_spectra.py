"""oracle/go_spectra.py -- CPU restatement of GfsOutputEnergySpectra (modules/fft.c:1340-1474) for a
uniform periodic box.  TEST INFRASTRUCTURE ONLY (see gfs_oracle.h).

The FFT arithmetic of the reference lives in FFTW3 (>= 3, with fftw3_mpi; un-vendored, absent here:
`modules/fft.h:26`, configure.ac:361-367) and the reference holds no test or golden data for its
spectra: PARITY UNPINNED.  What is restated is the published algorithm of the call sites:
  fill_cartesian_matrix (:966-1001)  v = (u - <u>)/ntot on the np^dim grid of the finest level
                                     (substract_average :897-908: <u> over the cells of all levels;
                                     get_data :399-417)
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


def all_level_mean(u):
    """substract_average (modules/fft.c:897-908): the volume-weighted average over the cells of ALL
    levels (FTT_TRAVERSE_ALL), the non-leaf values being the averages of their children
    (gfs_cell_coarse_init): sum_l h_l^dim sum_cells v / (number of levels).  u: the leaf level of one
    box, or of a whole lattice of boxes of `n_box' cells per side (levels below a box do not exist)."""
    return _all_level_mean(u, u.shape[0])


def _all_level_mean(u, n_box):
    dim = u.ndim
    val, vol = 0., 0.
    a = u
    m = n_box
    while True:
        h = 1. / m
        val += h ** dim * a.sum()
        vol += h ** dim * a.size
        if m == 1:
            break
        sh = []
        for s in a.shape:
            sh += [s // 2, 2]
        a = a.reshape(sh).mean(axis=tuple(range(1, 2 * dim, 2)))
        m //= 2
    return val / vol


def output_spectra(u, n_box=None):
    """GfsOutputSpectra (modules/fft.c:1101-1160) of a 3-D variable given as [k][j][i] (leaf level of
    the whole domain; n_box cells per side and box): the r2c DFT of (v - <v>)/ntot indexed
    [ix][iy][iz <= N/2], and the k step 2 pi/(x1 - x0) of write_spectra / init_kmax (:1031-1085)."""
    n = u.shape[0]
    nb = n if n_box is None else n_box
    a = np.transpose(u)
    a = (a - _all_level_mean(u, nb)) / float(a.size)
    dx = 1. / nb
    return np.fft.rfftn(a), 2. * np.pi / ((n - 1) * dx)


def output_spectra_plane(u, normal, pos):
    """GfsOutputSpectra with a flat box (realdim == 2, modules/fft.c:1131-1141): the points
    pos_min + j dx of the plane normal to `normal' (0 x, 1 y, 2 z) at coordinate pos take the value of the
    cell they lie in (fill_interpolated_cartesian_matrix, :822-883: gfs_domain_locate + GFS_VALUE), the mean
    of the np values is removed and the result divided by np; order_array (:800-820) puts the flat direction
    first and keeps the other two in coordinate order; fftw_plan_dft_r2c_3d (1, N, N).  u: [k][j][i] of the
    leaf level; returns F[ia][ib <= N/2] and the k step of write_spectra."""
    n = u.shape[0]
    kc = int(np.floor((pos + 0.5) * n))
    a = np.transpose(u)                      # [i][j][k]
    sl = [slice(None)] * 3
    sl[normal] = kc
    plane = a[tuple(sl)]                     # remaining axes in coordinate order
    npnt = plane.size
    avg = plane.sum() / npnt
    plane = (plane - avg) / npnt
    return np.fft.rfftn(plane), 2. * np.pi / ((n - 1) * (1. / n))


def turbulent_viscosity(u, Cs, model=1):
    """GfsVariableTurbulentViscosity (modules/turbulence.c:953-1048) on a uniform box: u = components
    WITH their ghost layer, [k][j][i]; returns the interior values.  g[i][j] = gfs_cm_gradient of
    component i along j for a variable that is not `centered' (src/fluid.c:3601-3615)."""
    dim = len(u)
    n = u[0].shape[0] - 2
    h = 1. / n
    inner = (slice(1, -1),) * dim

    def sh(a, axis, o):      # axis: 0 = x ... ; arrays are [k][j][i]
        sl = [slice(1, -1)] * dim
        ax = dim - 1 - axis
        sl[ax] = slice(1 + o, a.shape[ax] - 1 + o)
        return a[tuple(sl)]
    g = [[(sh(u[a], b, 1) - sh(u[a], b, -1) + (1. - 1.) * u[a][inner]) / 2. for b in range(dim)]
         for a in range(dim)]
    if model == 1:
        s = 0.
        for a in range(dim):
            for b in range(dim):
                e = 0.5 * (g[a][b] + g[b][a]) / h
                s = s + e * e
        return (Cs * h) * (Cs * h) * np.sqrt(2. * s)
    g2 = [[sum(g[m][a] * g[m][b] / (h * h) for m in range(dim)) for b in range(dim)] for a in range(dim)]
    g22 = [[sum(g2[a][m] * g2[m][b] for m in range(dim)) for b in range(dim)] for a in range(dim)]
    inv1 = sum(g2[a][a] for a in range(dim))
    inv2 = sum(g22[a][a] for a in range(dim))
    inv2 = (inv1 * inv1 - inv2) / 2
    if dim == 2:
        inv3 = g2[0][0] * g2[1][1] - g2[0][1] * g2[1][0]
    else:
        inv3 = (g2[0][0] * g2[1][1] * g2[2][2] + g2[0][1] * g2[1][2] * g2[2][0] + g2[0][2] * g2[1][0] * g2[2][1]
                - g2[0][2] * g2[1][1] * g2[2][0] - g2[0][1] * g2[1][0] * g2[2][2] - g2[0][0] * g2[1][2] * g2[2][1])
    alpha1 = inv1 * inv1 / 9. - inv2 / 3.
    alpha2 = inv1 ** 3 / 27. - inv1 * inv2 / 6. + inv3 / 2.
    with np.errstate(invalid="ignore", divide="ignore"):
        a32 = np.where(alpha1 > 0, alpha1, 1.) ** 1.5
        ok = (alpha1 > 0) & (alpha2 < a32)
        alpha3 = 1. / 3. * np.arccos(np.clip(alpha2 / a32, -1., 1.))
        sq = 2 * np.sqrt(np.where(alpha1 > 0, alpha1, 0.))
        s1 = np.sqrt(inv1 / 3. + sq * np.cos(alpha3))
        s2 = np.sqrt(inv1 / 3. - sq * np.cos(np.pi / 3. + alpha3))
        s3 = np.sqrt(inv1 / 3. - sq * np.cos(np.pi / 3. - alpha3))
        D = np.where(s1 != 0., s3 * (s1 - s2) * (s2 - s3) / np.where(s1 != 0., s1 * s1, 1.), 0.)
    return np.where(ok, (Cs * h) * (Cs * h) * D, 0.)


def energy_spectra(comps, n_box=None):
    """comps: interior arrays of the velocity components, indexed [k][j][i] (3-D) or [j][i] (2-D)
    as oracle.Field.interior() returns them (the whole domain; n_box cells per side and box when it
    is a lattice of boxes).  Returns (k, Ek, Etot): the printed lines and the total energy."""
    dim = comps[0].ndim
    n = comps[0].shape[0]
    nb = n if n_box is None else n_box
    nk = (dim + 1) * (n // 2 + 1) ** 2
    Ek = np.zeros(nk)
    kn = np.where(np.arange(n) < n // 2 + 1, np.arange(n), n - np.arange(n))
    for u in comps:
        a = np.transpose(u)                       # [ix][iy]([iz])
        a = (a - _all_level_mean(u, nb)) / float(n ** dim)
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
    dx = 1. / nb
    deltak = 2. * np.pi / ((n - 1) * dx)
    i = np.arange(1, nk)
    return deltak * np.sqrt(i.astype(float)), Ek[1:].copy(), Etot


def init_spectra(par, n):
    """GfsInitSpectra (modules/turbulence.c:545-901) restated with numpy for an n^3 box of unit size:
    par = dict(x0, y0, z0, L, E, alpha, epsilon, c1, c2, c3, ReL, kmax, seed, level).  Returns the
    three components indexed [k][j][i].  Same steps as the reference (see spectra.hip for the list);
    sums in numpy's order, the DFT is pocketfft's: agreement with the device to ~1e-12, not bit for
    bit.  rand() is the C library's (the reference calls srand (seed) before every rand())."""
    import ctypes
    libc = ctypes.CDLL(None)
    libc.srand(ctypes.c_uint(int(par["seed"])))
    c = 100. * (0.5 - libc.rand() / (2147483647. + 1.))
    m = 1 << par["level"]
    nh = m // 2 + 1
    deltak = 2. * np.pi / par["L"]
    sg = np.zeros((m, m, nh), dtype=np.int8)
    sg[:, :, 1:] = 1
    sg[1:nh, 1:, 0] = 1
    img = [(m - i, m - j) for i in range(1, nh) for j in range(1, m)]
    for j in range(1, nh):
        sg[0, j, 0] = 1
        sg[0, m - j, 0] = -1
    sg[1:nh, 0, 0] = 1
    img += [(m - i, 0) for i in range(1, nh)]
    for i, j in img:
        sg[i, j, 0] = -1
    idx = np.arange(m)
    kx = np.where(idx < nh, idx, idx - m) * deltak
    KX, KY, KZ = kx[:, None, None], kx[None, :, None], (np.arange(nh) * deltak)[None, None, :]
    k2 = KX ** 2 + KY ** 2 + KZ ** 2
    k2s = np.where(k2 != 0., k2, 1.)
    u = (1. + 1j * c * sg).astype(np.complex128)
    us = [(1. - KX ** 2 / k2s) * u - KX * KY / k2s * u - KX * KZ / k2s * u,
          - KY * KX / k2s * u + (1. - KY ** 2 / k2s) * u - KY * KZ / k2s * u,
          - KZ * KX / k2s * u - KY * KZ / k2s * u + (1. - KZ ** 2 / k2s) * u]
    us = [np.where(k2 != 0., a, 0.) for a in us]
    kn = np.where(idx < nh, idx, m - idx)
    bins = kn[:, None, None] ** 2 + kn[None, :, None] ** 2 + (np.arange(nh) ** 2)[None, None, :]
    w = np.where(np.arange(nh) == 0, 0.5, 1.)[None, None, :]
    nk = 3 * nh * nh
    Ek = np.zeros(nk)
    for a in us:
        np.add.at(Ek, bins.ravel(), (w * (a.real ** 2 + a.imag ** 2)).ravel())
    i = np.arange(nk)
    kwave = deltak * np.sqrt(i.astype(float))
    with np.errstate(divide="ignore", invalid="ignore"):
        if par["ReL"] != 0:
            Lint = par["E"] ** 1.5 / par["epsilon"]
            fl = (Lint * kwave / np.sqrt((Lint * kwave) ** 2 + par["c1"])) ** (11. / 3.)
            feta = np.exp(-par["c2"] * (((Lint * kwave * par["ReL"] ** (-0.75)) ** 4 + par["c3"] ** 4) ** 0.25
                                        - par["c3"]))
            Ei = par["alpha"] * par["epsilon"] ** (2. / 3.) * kwave ** (-5. / 3.) * fl * feta
        else:
            Ei = kwave ** 2
    Ei = np.where(kwave < par["kmax"], Ei, 0.)
    on = (Ek != 0.) & (i >= 1)
    Ei = np.where(on, Ei, 0.)
    cscale = np.where(on, np.sqrt(Ei / np.where(on, Ek, 1.)), 0.)
    cscale2 = np.sqrt(par["E"] / Ei.sum())
    out = []
    gx = [par[k] + par["L"] * (np.arange(m) / (m - 1.) - 0.5) for k in ("x0", "y0", "z0")]
    pc = (np.arange(n) + 0.5) / n - 0.5
    for a in us:
        a = a * (cscale2 * cscale[bins])
        g = np.fft.irfftn(a, s=(m, m, m), axes=(0, 1, 2)) * float(m) ** 3        # unnormalised c2r, [ix][iy][iz]
        # gfs_cartesian_grid_interpolate at the cell centres: z innermost, x outermost
        def interp(arr, axis, x, p):
            lo = np.clip(np.searchsorted(x, p, side="left") - 1, 0, len(x) - 2)
            v1, v2 = np.take(arr, lo, axis=axis), np.take(arr, lo + 1, axis=axis)
            shape = [1] * arr.ndim
            shape[axis] = len(p)
            t = ((p - x[lo]) / (x[lo + 1] - x[lo])).reshape(shape)
            return v1 + (v2 - v1) * t
        v = interp(g, 2, gx[2], pc)
        v = interp(v, 1, gx[1], pc)
        v = interp(v, 0, gx[0], pc)
        out.append(np.ascontiguousarray(np.transpose(v)))        # -> [k][j][i]
    return out
