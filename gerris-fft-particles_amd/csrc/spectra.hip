// spectra.hip -- GfsOutputEnergySpectra (modules/fft.c:1340-1474) on the device: hipFFT
// real-to-complex transforms of the velocity components of a uniform periodic box and the
// reference's binning of |F|^2 by integer |k|^2.
//
// The FFT arithmetic of the reference lives in FFTW3 (absent here) and the reference has no test or
// golden data for it: parity is unpinned; tests/ check this against a numpy restatement of the
// call sites (oracle/go_spectra.py) and against Parseval / single-mode properties.
//   fill_cartesian_matrix (:966-1001)   v = (u - <u>)/ntot           -> spectra_fill_kernel
//   get_fftw_plan (:1087-1098)           r2c DFT, last dimension halved -> hipfftExecD2Z
//   output_energy_spectra_event (:1360-1474) bins, weights             -> spectra_bin_kernel
// The dimensions are taken in the order x, y, z (z halved; the reference sorts them by size, all
// equal on a cube).  The bins are summed with fp64 atomics: the order of the additions, hence the
// last bits of Ek, differ from run to run and from the CPU loops (<= 1e-12 relative in the tests).
#include "gfship_internal.hpp"
#include <hipfft/hipfft.h>
#include <vector>
#include <cstdlib>
#include <cmath>

namespace gfship {

__global__ void __launch_bounds__(256)
spectra_sum_kernel (Layout L, const double * __restrict__ u, double * __restrict__ sum)
{
  // one block per (j, k) row
  const int n = L.n;
  const int j = 1 + blockIdx.x, k = L.dim == 3 ? 1 + blockIdx.y : 0;
  double s = 0.;
  for (int i = 1 + threadIdx.x; i <= n; i += blockDim.x)
    s += u[L.idx (i, j, k)];
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads ();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int) threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads ();
  }
  if (threadIdx.x == 0) atomicAdd (sum, sh[0]);
}

// a[ix][iy]([iz]) = (u - mean)/ntot
__global__ void __launch_bounds__(256)
spectra_fill_kernel (Layout L, const double * __restrict__ u, const double * __restrict__ sum,
		     double * __restrict__ a)
{
  const int n = L.n;
  const size_t ntot = L.dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const size_t q = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= ntot) return;
  const int i = (int) (q % n), j = (int) ((q/n) % n), k = L.dim == 3 ? (int) (q/((size_t) n*n)) : 0;
  const double mean = *sum/(double) ntot;
  const double v = (u[L.idx (i + 1, j + 1, L.dim == 3 ? k + 1 : 0)] - mean)/(double) ntot;
  const size_t dst = L.dim == 3 ? ((size_t) i*n + j)*n + k : (size_t) i*n + j;
  a[dst] = v;
}

__global__ void __launch_bounds__(256)
spectra_bin_kernel (int dim, int n, const double2 * __restrict__ F, double * __restrict__ Ek)
{
  const int nh = n/2 + 1;
  const size_t total = dim == 3 ? (size_t) n*n*nh : (size_t) n*nh;
  const size_t q = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= total) return;
  const double2 f = F[q];
  const double p = f.x*f.x + f.y*f.y;
  if (dim == 3) {
    const int k = (int) (q % nh), j = (int) ((q/nh) % n), i = (int) (q/((size_t) nh*n));
    const int knx = i < nh ? i : n - i, kny = j < nh ? j : n - j;
    atomicAdd (&Ek[knx*knx + kny*kny + k*k], k == 0 ? 0.5*p : p);
  }
  else {
    const int j = (int) (q % nh), i = (int) (q/nh);
    const int knx = i < nh ? i : n - i;
    if (j == 0) atomicAdd (&Ek[knx*knx], 0.5*p);
    atomicAdd (&Ek[knx*knx + j*j], p);           /* j = 0 again, as in the reference (:1428-1432) */
  }
}

// get_cell_values (modules/turbulence.c:374-389) = gfs_cartesian_grid_interpolate (src/cartesian.c:
// 205-257) of the np^3 grid at the cell centres: nested 1-D interpolations, x outermost
__device__ __forceinline__ int cgd_lookup (const double * __restrict__ x, int n, double p)
{
  unsigned min = 0, max = n - 1;
  if (p < x[min] || p > x[max])
    return -1;
  while (max > min + 1) {
    unsigned m = (min + max)/2;
    if (p > x[m]) min = m; else max = m;
  }
  return (int) min;
}

__global__ void __launch_bounds__(256)
spectra_cells_kernel (Layout L, int np, const double * __restrict__ g, const double * __restrict__ gx,
		      const double * __restrict__ gy, const double * __restrict__ gz, double Lbox,
		      double * __restrict__ u, unsigned * __restrict__ outside)
{
  const int n = L.n;
  const size_t q = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= (size_t) n*n*n) return;
  const int ci = (int) (q % n) + 1, cj = (int) ((q/n) % n) + 1, ck = (int) (q/((size_t) n*n)) + 1;
  const double h = 1./n;
  const double p[3] = { -0.5 + (ci - 0.5)*h, -0.5 + (cj - 0.5)*h, -0.5 + (ck - 0.5)*h };
  const int i = cgd_lookup (gx, np, p[0]), j = cgd_lookup (gy, np, p[1]), k = cgd_lookup (gz, np, p[2]);
  if (i < 0 || j < 0 || k < 0) {
    atomicAdd (outside, 1u);
    u[L.idx (ci, cj, ck)] = 0.;
    return;
  }
  double vx[2];
#pragma unroll
  for (int a = 0; a < 2; a++) {
    double vy[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const double * row = g + ((size_t) (i + a)*np + (j + b))*np;
      const double v1 = row[k]/Lbox, v2 = row[k + 1]/Lbox;
      vy[b] = v1 + (v2 - v1)*(p[2] - gz[k])/(gz[k + 1] - gz[k]);
    }
    vx[a] = vy[0] + (vy[1] - vy[0])*(p[1] - gy[j])/(gy[j + 1] - gy[j]);
  }
  u[L.idx (ci, cj, ck)] = vx[0] + (vx[1] - vx[0])*(p[0] - gx[i])/(gx[i + 1] - gx[i]);
}

} // namespace gfship

using namespace gfship;

#define GFSHIP_FFT(call) do { hipfftResult r_ = (call); if (r_ != HIPFFT_SUCCESS) {	\
      gfship::set_error ("hipFFT error %d in %s", (int) r_, #call); return GFSHIP_EHIP; } } while (0)

extern "C" {

int gfship_energy_spectra_bins (gfship_domain * dom)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  int nh = dom->lay[dom->depth].n/2 + 1;
  return (dom->dim + 1)*nh*nh;
}

int gfship_energy_spectra (gfship_domain * dom, int ncomp, const gfship_field * comps, double * Ek,
			   double * Etot, double * deltak)
{
  GFSHIP_CHECK (dom && comps && Ek && ncomp > 0, GFSHIP_EINVAL, "invalid argument");
  for (int d = 0; d < 2*dom->dim; d++)
    GFSHIP_CHECK (dom->side[d] == GFSHIP_SIDE_PERIODIC || dom->side[d] == GFSHIP_SIDE_BOUNDARY, GFSHIP_EINVAL,
		  "spectra of a multi-box domain are not supported");
  const Layout & L = dom->lay[dom->depth];
  const int n = L.n, dim = dom->dim, nh = n/2 + 1;
  GFSHIP_CHECK (n >= 2, GFSHIP_EINVAL, "level too coarse for a spectrum");
  const size_t ntot = dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const size_t nout = dim == 3 ? (size_t) n*n*nh : (size_t) n*nh;
  const int nk = gfship_energy_spectra_bins (dom);
  double * a = nullptr, * dEk = nullptr, * sum = nullptr;
  double2 * F = nullptr;
  GFSHIP_HIP (hipMalloc ((void **) &a, ntot*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &F, nout*sizeof (double2)));
  GFSHIP_HIP (hipMalloc ((void **) &dEk, (size_t) nk*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &sum, sizeof (double)));
  GFSHIP_HIP (hipMemsetAsync (dEk, 0, (size_t) nk*sizeof (double), dom->stream));
  hipfftHandle plan;
  if (dim == 3) GFSHIP_FFT (hipfftPlan3d (&plan, n, n, n, HIPFFT_D2Z));
  else GFSHIP_FFT (hipfftPlan2d (&plan, n, n, HIPFFT_D2Z));
  GFSHIP_FFT (hipfftSetStream (plan, dom->stream));
  for (int c = 0; c < ncomp; c++) {
    Field * fc = get_field (dom, comps[c]);
    GFSHIP_CHECK (fc != nullptr, GFSHIP_EINVAL, "invalid field");
    const double * u = fc->lev[dom->depth];
    GFSHIP_HIP (hipMemsetAsync (sum, 0, sizeof (double), dom->stream));
    hipLaunchKernelGGL (spectra_sum_kernel, dim3 (n, dim == 3 ? n : 1), dim3 (256), 0, dom->stream, L, u, sum);
    hipLaunchKernelGGL (spectra_fill_kernel, dim3 ((unsigned) ((ntot + 255)/256)), dim3 (256), 0, dom->stream,
			L, u, sum, a);
    GFSHIP_HIP (hipGetLastError ());
    GFSHIP_FFT (hipfftExecD2Z (plan, a, (hipfftDoubleComplex *) F));
    hipLaunchKernelGGL (spectra_bin_kernel, dim3 ((unsigned) ((nout + 255)/256)), dim3 (256), 0, dom->stream,
			dim, n, F, dEk);
    GFSHIP_HIP (hipGetLastError ());
  }
  GFSHIP_HIP (hipMemcpyAsync (Ek, dEk, (size_t) nk*sizeof (double), hipMemcpyDeviceToHost, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  (void) hipfftDestroy (plan);
  (void) hipFree (a); (void) hipFree (F); (void) hipFree (dEk); (void) hipFree (sum);
  if (Etot) {
    double e = 0.;
    for (int q = 0; q < nk; q++) e += Ek[q];
    *Etot = e;
  }
  if (deltak) {
    /* x1 - x0 = distance between the first and the last cell centres (get_deep_level, :443-450) */
    double dx = 1./n;
    *deltak = 2.*M_PI/((n - 1)*dx);
  }
  return GFSHIP_OK;
}

// GfsInitSpectra (modules/turbulence.c:545-901): synthetic velocity field with a prescribed energy
// spectrum.  Steps of gfs_init_spectra_event (:747-901), one process:
//   generate_vel_field (:545-617)     every Fourier mode (1, +-c) for the three components, with
//       c = 100 (0.5 - rand()/(RAND_MAX + 1)) and srand (seed) called before EVERY rand(): the same
//       c for all modes (as in the reference); Hermitian images on the kz = 0 plane set afterwards
//   solenoidal_vel_field (:620-665)   u - k (k.u)/k^2
//   spectral_energy (:672-745)        shell energies, target E(k) (ReL != 0: Pope's model spectrum
//       alpha eps^2/3 k^-5/3 fL feta; ReL = 0: k^2) below kmax, rescaling to the total energy E
//   complex-to-real DFT of each component (hipFFT here, FFTW there) / Lbox onto the np^3 grid with
//       points at pos + L (j/(np - 1) - 0.5), interpolated at the cell centres (get_cell_values)
// The host loops below run in the reference's order with the same libm calls; only the DFT differs.
int gfship_init_spectra (gfship_domain * dom, const gfship_init_spectra_params * par,
			 const gfship_field v[3])
{
  GFSHIP_CHECK (dom && par && v, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (dom->dim == 3, GFSHIP_EUNSUPPORTED, "GfsInitSpectra only works in 3-D (turbulence.c:747)");
  GFSHIP_CHECK (par->level >= 1 && par->level <= 9, GFSHIP_EINVAL, "level of the spectral grid out of range");
  const int np = 1 << par->level, nh = np/2 + 1;
  const double deltak = 2.*M_PI/par->L;
  const size_t nc = (size_t) np*np*nh;
  auto index = [np, nh] (int i, int j, int k) { return (size_t) k + (size_t) nh*((size_t) j + (size_t) np*i); };

  // ---- generate_vel_field ----
  srand ((unsigned) par->seed);
  const double c = 100.*(0.5 - (rand ()/((double) RAND_MAX + 1)));
  std::vector<signed char> sg (nc, 0);          // sign of the imaginary part, 0 = never set
  for (int i = 0; i < np; i++)
    for (int j = 0; j < np; j++)
      for (int k = 1; k < nh; k++)
	sg[index (i, j, k)] = 1;
  std::vector<size_t> images;
  for (int i = 1; i < nh; i++)
    for (int j = 1; j < np; j++) {
      sg[index (i, j, 0)] = 1;
      images.push_back (index (np - i, np - j, 0));
    }
  for (int j = 1; j < nh; j++) {
    sg[index (0, j, 0)] = 1;
    sg[index (0, np - j, 0)] = -1;
  }
  for (int i = 1; i < nh; i++) {
    sg[index (i, 0, 0)] = 1;
    images.push_back (index (np - i, 0, 0));
  }
  for (size_t q : images) sg[q] = -1;           /* assign_data_parallel */

  // ---- solenoidal_vel_field ----
  std::vector<double> us[3];
  for (int d = 0; d < 3; d++) us[d].assign (2*nc, 0.);
  for (int i = 0; i < np; i++) {
    double kx = i < nh ? i*deltak : (i - np)*deltak;
    for (int j = 0; j < np; j++) {
      double ky = j < nh ? j*deltak : (j - np)*deltak;
      for (int k = 0; k < nh; k++) {
	double kz = k*deltak;
	double kmod2 = pow (kx, 2) + pow (ky, 2) + pow (kz, 2);
	size_t q = index (i, j, k);
	if (kmod2 != 0.) {
	  const double ur = 1., ui = sg[q]*c;       /* the same for the three components */
	  for (int part = 0; part < 2; part++) {
	    const double u0 = part ? ui : ur, u1 = u0, u2 = u0;
	    us[0][2*q + part] = (1. - pow (kx, 2)/kmod2)*u0 - kx*ky/kmod2*u1 - kx*kz/kmod2*u2;
	    us[1][2*q + part] = - ky*kx/kmod2*u0 + (1. - pow (ky, 2)/kmod2)*u1 - ky*kz/kmod2*u2;
	    us[2][2*q + part] = - kz*kx/kmod2*u0 - ky*kz/kmod2*u1 + (1. - pow (kz, 2)/kmod2)*u2;
	  }
	}
      }
    }
  }

  // ---- spectral_energy ----
  const int nk = 3*nh*nh;
  std::vector<double> Ek (nk, 0.), cscale (nk, 0.);
  for (int d = 0; d < 3; d++)
    for (int i = 0; i < np; i++) {
      int knx = i < nh ? i : np - i;
      for (int j = 0; j < np; j++) {
	int kny = j < nh ? j : np - j;
	size_t q = index (i, j, 0);
	Ek[knx*knx + kny*kny] += 0.5*(pow (us[d][2*q], 2) + pow (us[d][2*q + 1], 2));
	for (int k = 1; k < nh; k++) {
	  q = index (i, j, k);
	  Ek[knx*knx + kny*kny + k*k] += pow (us[d][2*q], 2) + pow (us[d][2*q + 1], 2);
	}
      }
    }
  double Ekspectra = 0.;
  const double Lint = pow (par->E, 3./2.)/par->epsilon;
  for (int i = 1; i < nk; i++) {
    double kwave = deltak*sqrt ((double) i), Ei;
    if (Ek[i] != 0.) {
      if (par->ReL != 0) {
	double fl = pow (Lint*kwave/sqrt (pow (Lint*kwave, 2) + par->c1), 11./3.);
	double feta = exp (- par->c2*(pow (pow (Lint*kwave*pow (par->ReL, -3./4.), 4) + pow (par->c3, 4), 0.25) - par->c3));
	if (kwave < par->kmax)
	  Ei = par->alpha*pow (par->epsilon, 2./3.)*pow (kwave, -5./3.)*fl*feta;
	else
	  Ei = 0.;
      }
      else {
	if (kwave < par->kmax)
	  Ei = pow (kwave, 2.);
	else
	  Ei = 0.;
      }
      cscale[i] = sqrt (Ei/Ek[i]);
      Ekspectra += Ei;
    }
  }
  GFSHIP_CHECK (Ekspectra > 0., GFSHIP_EINVAL, "the target spectrum is empty (kmax below the first shell?)");
  const double cscale2 = sqrt (par->E/Ekspectra);
  for (int i = 0; i < np; i++) {
    int knx = i < nh ? i : np - i;
    for (int j = 0; j < np; j++) {
      int kny = j < nh ? j : np - j;
      for (int k = 0; k < nh; k++) {
	int b = knx*knx + kny*kny + k*k;
	size_t q = index (i, j, k);
	for (int d = 0; d < 3; d++) {
	  us[d][2*q] *= cscale2*cscale[b];
	  us[d][2*q + 1] *= cscale2*cscale[b];
	}
      }
    }
  }

  // ---- complex-to-real transforms and the cells ----
  std::vector<double> gx[3];
  const double pos[3] = { par->x0, par->y0, par->z0 };
  for (int a = 0; a < 3; a++) {
    gx[a].resize (np);
    for (int j = 0; j < np; j++)
      gx[a][j] = pos[a] + par->L*((double) j/(np - 1) - 0.5);
  }
  const Layout & L = dom->lay[dom->depth];
  double2 * F = nullptr;
  double * g = nullptr, * dgx = nullptr;
  unsigned * outside = nullptr;
  GFSHIP_HIP (hipMalloc ((void **) &F, nc*sizeof (double2)));
  GFSHIP_HIP (hipMalloc ((void **) &g, (size_t) np*np*np*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &dgx, 3*(size_t) np*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &outside, sizeof (unsigned)));
  GFSHIP_HIP (hipMemsetAsync (outside, 0, sizeof (unsigned), dom->stream));
  for (int a = 0; a < 3; a++)
    GFSHIP_HIP (hipMemcpyAsync (dgx + (size_t) a*np, gx[a].data (), np*sizeof (double), hipMemcpyHostToDevice,
				dom->stream));
  hipfftHandle plan;
  GFSHIP_FFT (hipfftPlan3d (&plan, np, np, np, HIPFFT_Z2D));
  GFSHIP_FFT (hipfftSetStream (plan, dom->stream));
  const size_t ncells = (size_t) L.n*L.n*L.n;
  for (int d = 0; d < 3; d++) {
    Field * fd = get_field (dom, v[d]);
    GFSHIP_CHECK (fd != nullptr, GFSHIP_EINVAL, "invalid field");
    GFSHIP_HIP (hipMemcpyAsync (F, us[d].data (), nc*sizeof (double2), hipMemcpyHostToDevice, dom->stream));
    GFSHIP_FFT (hipfftExecZ2D (plan, (hipfftDoubleComplex *) F, g));
    hipLaunchKernelGGL (spectra_cells_kernel, dim3 ((unsigned) ((ncells + 255)/256)), dim3 (256), 0,
			dom->stream, L, np, g, dgx, dgx + np, dgx + 2*(size_t) np, 1., fd->lev[dom->depth],
			outside);
    GFSHIP_HIP (hipGetLastError ());
    fd->zero[dom->depth] = false;
    GFSHIP_HIP (hipStreamSynchronize (dom->stream));      /* us[d] is read by the copy */
  }
  unsigned nout = 0;
  GFSHIP_HIP (hipMemcpy (&nout, outside, sizeof (unsigned), hipMemcpyDeviceToHost));
  (void) hipfftDestroy (plan);
  (void) hipFree (F); (void) hipFree (g); (void) hipFree (dgx); (void) hipFree (outside);
  GFSHIP_CHECK (nout == 0, GFSHIP_EINVAL,
		"%u cells lie outside the spectral grid (the reference leaves them undefined)", nout);
  return GFSHIP_OK;
}

} // extern "C"
