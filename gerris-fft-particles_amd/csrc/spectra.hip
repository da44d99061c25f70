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

} // extern "C"
