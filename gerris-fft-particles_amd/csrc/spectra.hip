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

// substract_average (modules/fft.c:897-908): the average is taken over the cells of ALL levels
// (FTT_TRAVERSE_ALL) weighted by their volumes: sum_l h_l^dim sum_cells v / (number of levels)
__global__ void spectra_mean_kernel (int dim, int depth, const double * __restrict__ sums,
				     double * __restrict__ mean, double nboxes)
{
  if (threadIdx.x || blockIdx.x) return;
  double val = 0., vol = 0.;
  for (int l = 0; l <= depth; l++) {
    const double h = 1./(double) (1 << l);
    const double cv = dim == 3 ? h*h*h : h*h;
    val += cv*sums[l];
    vol += nboxes*cv*(double) (dim == 3 ? (1ull << (3*l)) : (1ull << (2*l)));
  }
  *mean = val/vol;
}

// get_data (modules/fft.c:399-417): a[ix][iy]([iz]) = (u - mean)/ntot on the grid of the whole
// lattice of boxes (N = b n points per side; this box at offset o)
__global__ void __launch_bounds__(256)
spectra_fill_kernel (Layout L, const double * __restrict__ u, const double * __restrict__ mean,
		     double * __restrict__ a, double ntot_all)
{
  const int n = L.n;
  const size_t ntot = L.dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const size_t q = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= ntot) return;
  const int i = (int) (q % n), j = (int) ((q/n) % n), k = L.dim == 3 ? (int) (q/((size_t) n*n)) : 0;
  const double v = (u[L.idx (i + 1, j + 1, L.dim == 3 ? k + 1 : 0)] - *mean)/ntot_all;
  const size_t dst = L.dim == 3 ? ((size_t) i*n + j)*n + k : (size_t) i*n + j;
  a[dst] = v;
}

// the blocks of the boxes of a lattice (gathered in rank order, each [ix][iy][iz]) into the grid of
// the whole domain: what the slab redistribution of the reference (modules/fft.c:467-669) assembles
__global__ void __launch_bounds__(256)
spectra_assemble_kernel (int dim, int n, int bx, int by, int bz, const double * __restrict__ g,
			 double * __restrict__ a)
{
  const size_t nloc = dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const size_t total = nloc*bx*by*bz;
  const size_t q = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= total) return;
  const int box = (int) (q/nloc);
  const size_t r = q % nloc;
  const int cx = box % bx, cy = (box/bx) % by, cz = box/(bx*by);
  const int Ny = by*n, Nz = bz*n;
  if (dim == 3) {
    const int k = (int) (r % n), j = (int) ((r/n) % n), i = (int) (r/((size_t) n*n));
    a[((size_t) (cx*n + i)*Ny + (cy*n + j))*Nz + cz*n + k] = g[q];
  }
  else {
    const int j = (int) (r % n), i = (int) (r/n);
    a[(size_t) (cx*n + i)*Ny + cy*n + j] = g[q];
  }
}

__global__ void __launch_bounds__(256)
spectra_bin_kernel (int dim, int n, const double2 * __restrict__ F, double * __restrict__ Ek)
{
  // n: points per side of the (cubic) grid
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

// GfsVariableTurbulentViscosity (modules/turbulence.c:953-1062): eddy viscosity of the leaf cells from
// the velocity gradients g[i][j] = gfs_cm_gradient of component i along j (src/fluid.c:3586-3617, the
// branch of a variable that is not `centered': (g1.b - g2.b + (g2.a - g1.a)*val)/2. with g.a = 1.,
// g.b = neighbour value).  model 1: Smagorinsky (:961-978, the only one a file can select: the read
// method never changes model_type = 1); model 0: the sigma model of :980-1048.
template <int DIM>
__global__ void __launch_bounds__(256)
turbulent_viscosity_kernel (Layout L, const double * __restrict__ u0, const double * __restrict__ u1,
			    const double * __restrict__ u2, double Cs, int model, double * __restrict__ out)
{
  const int n = L.n;
  const size_t ntot = DIM == 3 ? (size_t) n*n*n : (size_t) n*n;
  const size_t q = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= ntot) return;
  const int i = (int) (q % n) + 1, j = (int) ((q/n) % n) + 1, k = DIM == 3 ? (int) (q/((size_t) n*n)) + 1 : 0;
  const long c = L.idx (i, j, k);
  const long off[3] = { 1, L.sy, L.sz };
  const double * u[3] = { u0, u1, u2 };
  const double h = 1./n;
  double g[3][3];
#pragma unroll
  for (int a = 0; a < DIM; a++)
#pragma unroll
    for (int b = 0; b < DIM; b++) {
      const double val = u[a][c];
      g[a][b] = (u[a][c + off[b]] - u[a][c - off[b]] + (1. - 1.)*val)/2.;
    }
  if (model == 1) {
    double s = 0.;
#pragma unroll
    for (int a = 0; a < DIM; a++)
#pragma unroll
      for (int b = 0; b < DIM; b++) {
	const double e = 0.5*(g[a][b] + g[b][a])/h;
	s += e*e;                                       /* pow (x, 2) */
      }
    s = sqrt (2.*s);
    out[c] = (Cs*h)*(Cs*h)*s;
    return;
  }
  double g2[3][3], g22[3][3];
#pragma unroll
  for (int a = 0; a < DIM; a++)
#pragma unroll
    for (int b = 0; b < DIM; b++) {
      g2[a][b] = 0.;
      for (int m = 0; m < DIM; m++)
	g2[a][b] += g[m][a]*g[m][b]/(h*h);
    }
#pragma unroll
  for (int a = 0; a < DIM; a++)
#pragma unroll
    for (int b = 0; b < DIM; b++) {
      g22[a][b] = 0.;
      for (int m = 0; m < DIM; m++)
	g22[a][b] += g2[a][m]*g2[m][b];
    }
  double inv1 = 0., inv2 = 0., inv3;
  for (int a = 0; a < DIM; a++) { inv1 += g2[a][a]; inv2 += g22[a][a]; }
  inv2 = (inv1*inv1 - inv2)/2;
  if (DIM == 2)
    inv3 = g2[0][0]*g2[1][1] - g2[0][1]*g2[1][0];
  else {
    inv3  = g2[0][0]*g2[1][1]*g2[2][2];
    inv3 += g2[0][1]*g2[1][2]*g2[2][0];
    inv3 += g2[0][2]*g2[1][0]*g2[2][1];
    inv3 -= g2[0][2]*g2[1][1]*g2[2][0];
    inv3 -= g2[0][1]*g2[1][0]*g2[2][2];
    inv3 -= g2[0][0]*g2[1][2]*g2[2][1];
  }
  const double alpha1 = inv1*inv1/9. - inv2/3.;
  const double alpha2 = inv1*inv1*inv1/27. - inv1*inv2/6. + inv3/2.;
  if (alpha1 <= 0) { out[c] = 0.; return; }
  const double a32 = pow (alpha1, 3./2.);
  if (alpha2 >= a32) { out[c] = 0.; return; }
  const double alpha3 = 1./3.*acos (alpha2/a32);
  const double sigma1 = sqrt (inv1/3. + 2*sqrt (alpha1)*cos (alpha3));
  const double sigma2 = sqrt (inv1/3. - 2*sqrt (alpha1)*cos (M_PI/3. + alpha3));
  const double sigma3 = sqrt (inv1/3. - 2*sqrt (alpha1)*cos (M_PI/3. - alpha3));
  double Dsigma;
  if (sigma1 != 0.)
    Dsigma = sigma3*(sigma1 - sigma2)*(sigma2 - sigma3)/(sigma1*sigma1);
  else
    Dsigma = 0.;
  out[c] = (Cs*h)*(Cs*h)*Dsigma;
}

// GfsOutputSpectra of a plane (modules/fft.c:1101-1160 with realdim == 2: fill_interpolated_cartesian_matrix,
// :822-883): the values of the cells the points pos_min + j dx of the plane lie in (gfs_domain_locate, no
// interpolation), minus their mean, divided by their number.  Plane normal to `normal' through the cells of
// index kc along it; a = first, b = second in-plane coordinate (in coordinate order)
__global__ void __launch_bounds__(256)
spectra_plane_gather_kernel (Layout L, const double * __restrict__ u, int normal, int kc, double * __restrict__ a,
			     double * __restrict__ sum)
{
  const int n = L.n;
  const int q = blockIdx.x*blockDim.x + threadIdx.x;
  double v = 0.;
  if (q < n*n) {
    const int ib = q % n + 1, ia = q / n + 1;
    int ijk[3];
    ijk[normal] = kc;
    ijk[normal == 0 ? 1 : 0] = ia;
    ijk[normal == 2 ? 1 : 2] = ib;
    v = u[L.idx (ijk[0], ijk[1], ijk[2])];
    a[q] = v;
  }
  __shared__ double sh[256];
  sh[threadIdx.x] = v;
  __syncthreads ();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int) threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads ();
  }
  if (threadIdx.x == 0) atomicAdd (sum, sh[0]);
}

__global__ void __launch_bounds__(256)
spectra_plane_center_kernel (int np, const double * __restrict__ sum, double * __restrict__ a)
{
  const int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= np) return;
  const double avg = *sum/np;
  double v = a[q];
  v -= avg;
  v /= np;
  a[q] = v;
}

} // namespace gfship

using namespace gfship;

#define GFSHIP_FFT(call) do { hipfftResult r_ = (call); if (r_ != HIPFFT_SUCCESS) {	\
      gfship::set_error ("hipFFT error %d in %s", (int) r_, #call); return GFSHIP_EHIP; } } while (0)

extern "C" {

// points per side of the grid of the whole domain; 0 when the lattice of boxes is not a cube
static int spectra_side (const gfship_domain * dom)
{
  const int n = dom->lay[dom->depth].n;
  if (!dom->has_external) return n;
  for (int c = 1; c < dom->dim; c++)
    if (dom->lat_b[c] != dom->lat_b[0]) return 0;
  return n*dom->lat_b[0];
}

int gfship_energy_spectra_bins (gfship_domain * dom)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  int nh = spectra_side (dom)/2 + 1;
  return (dom->dim + 1)*nh*nh;
}

namespace {

// fill_cartesian_matrix + get_fftw_plan + fftw_execute for one variable: F = r2c DFT of
// (v - <v>)/ntot on the grid of the whole domain.  On a box of a lattice every rank gathers the
// blocks of all boxes and transforms the whole grid itself (replicated; the reference redistributes
// slabs for FFTW-MPI).
struct SpectraWork {
  gfship_domain * dom = nullptr;
  int N = 0;
  size_t ntot = 0, nout = 0;
  double * a = nullptr, * loc = nullptr, * gat = nullptr, * sums = nullptr, * mean = nullptr;
  double2 * F = nullptr;
  hipfftHandle plan = 0;
  bool have_plan = false;
  ~SpectraWork () {
    if (have_plan) (void) hipfftDestroy (plan);
    (void) hipFree (a); (void) hipFree (loc); (void) hipFree (gat); (void) hipFree (sums);
    (void) hipFree (mean); (void) hipFree (F);
  }
};

int spectra_prepare (gfship_domain * dom, SpectraWork * W)
{
  for (int d = 0; d < 2*dom->dim; d++)
    GFSHIP_CHECK (dom->side[d] != GFSHIP_SIDE_EXTERNAL || dom->comm || dom->gather, GFSHIP_EINVAL,
		  "spectra of a box of a lattice need the communicator or the gather hook");
  const int dim = dom->dim;
  W->dom = dom;
  W->N = spectra_side (dom);
  GFSHIP_CHECK (W->N > 0, GFSHIP_EUNSUPPORTED, "the lattice of boxes must have the same extent in every direction");
  GFSHIP_CHECK (W->N >= 2, GFSHIP_EINVAL, "level too coarse for a spectrum");
  const int N = W->N, nh = N/2 + 1;
  W->ntot = dim == 3 ? (size_t) N*N*N : (size_t) N*N;
  W->nout = dim == 3 ? (size_t) N*N*nh : (size_t) N*nh;
  GFSHIP_HIP (hipMalloc ((void **) &W->a, W->ntot*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &W->F, W->nout*sizeof (double2)));
  GFSHIP_HIP (hipMalloc ((void **) &W->sums, (GFSHIP_MAXLEVEL + 2)*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &W->mean, sizeof (double)));
  if (dom->has_external) {
    const int n = dom->lay[dom->depth].n;
    const size_t nloc = dim == 3 ? (size_t) n*n*n : (size_t) n*n;
    GFSHIP_HIP (hipMalloc ((void **) &W->loc, nloc*sizeof (double)));
    GFSHIP_HIP (hipMalloc ((void **) &W->gat, W->ntot*sizeof (double)));
  }
  if (dim == 3) GFSHIP_FFT (hipfftPlan3d (&W->plan, N, N, N, HIPFFT_D2Z));
  else GFSHIP_FFT (hipfftPlan2d (&W->plan, N, N, HIPFFT_D2Z));
  W->have_plan = true;
  GFSHIP_FFT (hipfftSetStream (W->plan, dom->stream));
  return GFSHIP_OK;
}

int spectra_transform (SpectraWork * W, gfship_field v)
{
  gfship_domain * dom = W->dom;
  Field * fc = get_field (dom, v);
  GFSHIP_CHECK (fc != nullptr, GFSHIP_EINVAL, "invalid field");
  const int dim = dom->dim, depth = dom->depth;
  const Layout & L = dom->lay[depth];
  const int n = L.n;
  const size_t nloc = dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  /* the non-leaf values are the ones gfs_cell_coarse_init last gave the variable (for the variables
     of a simulation: the state before the last approximate projection, as in the reference, whose
     events run at the top of the loop); a field that never had any gets them from its leaves */
  if (!fc->coarse_valid && depth > 0) {
    Field * one[1] = { fc };
    if (int r = launch_coarse_init (dom, one, 1)) return r;
  }
  // substract_average over the cells of all levels
  GFSHIP_HIP (hipMemsetAsync (W->sums, 0, (GFSHIP_MAXLEVEL + 2)*sizeof (double), dom->stream));
  for (int l = 0; l <= depth; l++) {
    const Layout & Ll = dom->lay[l];
    hipLaunchKernelGGL (spectra_sum_kernel, dim3 (Ll.n, dim == 3 ? Ll.n : 1), dim3 (256), 0, dom->stream,
			Ll, (const double *) fc->lev[l], W->sums + l);
  }
  GFSHIP_HIP (hipGetLastError ());
  double nboxes = 1.;
  if (dom->has_external) {
    /* gfs_all_reduce of the sums (modules/fft.c:902-903) */
    std::vector<double> h (depth + 1);
    GFSHIP_HIP (hipMemcpyAsync (h.data (), W->sums, (depth + 1)*sizeof (double), hipMemcpyDeviceToHost, dom->stream));
    GFSHIP_HIP (hipStreamSynchronize (dom->stream));
    for (int l0 = 0; l0 <= depth; l0 += 8) {
      int m = depth + 1 - l0 < 8 ? depth + 1 - l0 : 8;
      if (int r = call_reduce (dom, h.data () + l0, m, 0)) return r;
    }
    GFSHIP_HIP (hipMemcpyAsync (W->sums, h.data (), (depth + 1)*sizeof (double), hipMemcpyHostToDevice, dom->stream));
    GFSHIP_HIP (hipStreamSynchronize (dom->stream));
    nboxes = (double) dom->lat_n;
  }
  hipLaunchKernelGGL (spectra_mean_kernel, dim3 (1), dim3 (64), 0, dom->stream, dim, depth, W->sums, W->mean, nboxes);
  double * dst = dom->has_external ? W->loc : W->a;
  hipLaunchKernelGGL (spectra_fill_kernel, dim3 ((unsigned) ((nloc + 255)/256)), dim3 (256), 0, dom->stream,
		      L, (const double *) fc->lev[depth], W->mean, dst, (double) W->ntot);
  GFSHIP_HIP (hipGetLastError ());
  if (dom->has_external) {
    if (int r = call_gather (dom, W->loc, W->gat, nloc)) return r;
    hipLaunchKernelGGL (spectra_assemble_kernel, dim3 ((unsigned) ((W->ntot + 255)/256)), dim3 (256), 0, dom->stream,
			dim, n, dom->lat_b[0], dom->lat_b[1], dom->lat_b[2], W->gat, W->a);
    GFSHIP_HIP (hipGetLastError ());
  }
  GFSHIP_FFT (hipfftExecD2Z (W->plan, W->a, (hipfftDoubleComplex *) W->F));
  return GFSHIP_OK;
}

} // namespace

int gfship_energy_spectra (gfship_domain * dom, int ncomp, const gfship_field * comps, double * Ek,
			   double * Etot, double * deltak)
{
  GFSHIP_CHECK (dom && comps && Ek && ncomp > 0, GFSHIP_EINVAL, "invalid argument");
  SpectraWork W;
  int r = spectra_prepare (dom, &W);
  if (r) return r;
  const int dim = dom->dim, N = W.N;
  const int nk = gfship_energy_spectra_bins (dom);
  double * dEk = nullptr;
  GFSHIP_HIP (hipMalloc ((void **) &dEk, (size_t) nk*sizeof (double)));
  hipError_t e = hipMemsetAsync (dEk, 0, (size_t) nk*sizeof (double), dom->stream);
  for (int c = 0; c < ncomp && e == hipSuccess && r == GFSHIP_OK; c++) {
    r = spectra_transform (&W, comps[c]);
    if (r) break;
    hipLaunchKernelGGL (spectra_bin_kernel, dim3 ((unsigned) ((W.nout + 255)/256)), dim3 (256), 0, dom->stream,
			dim, N, W.F, dEk);
    e = hipGetLastError ();
  }
  if (r == GFSHIP_OK && e == hipSuccess)
    e = hipMemcpyAsync (Ek, dEk, (size_t) nk*sizeof (double), hipMemcpyDeviceToHost, dom->stream);
  if (r == GFSHIP_OK && e == hipSuccess)
    e = hipStreamSynchronize (dom->stream);
  (void) hipFree (dEk);
  if (r) return r;
  GFSHIP_HIP (e);
  if (Etot) {
    double s = 0.;
    for (int q = 0; q < nk; q++) s += Ek[q];
    *Etot = s;
  }
  if (deltak) {
    /* x1 - x0 = distance between the first and the last cell centres (get_deep_level, :443-450) */
    double dx = 1./dom->lay[dom->depth].n;
    *deltak = 2.*M_PI/((N - 1)*dx);
  }
  return GFSHIP_OK;
}

// GfsOutputSpectra (modules/fft.c:1101-1160) of the whole domain at the finest level, 3-D: the r2c
// DFT of (v - <v>)/ntot; write_spectra (:1047-1085) prints, for ix, iy = 0 .. N-1 and iz = 0 .. N/2,
// "kx ky kz re*L im*L" with k = 2 pi/(x1 - x0) times the signed index.
int gfship_output_spectra_side (gfship_domain * dom)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  return spectra_side (dom);
}

int gfship_output_spectra (gfship_domain * dom, gfship_field v, double * out, double * kstep)
{
  GFSHIP_CHECK (dom && out, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (dom->dim == 3, GFSHIP_EUNSUPPORTED,
		"GfsOutputSpectra of a plane interpolates through the tree (fill_interpolated_cartesian_matrix): "
		"only the 3-D box is transformed here");
  SpectraWork W;
  int r = spectra_prepare (dom, &W);
  if (r) return r;
  if ((r = spectra_transform (&W, v))) return r;
  GFSHIP_HIP (hipMemcpyAsync (out, W.F, W.nout*sizeof (double2), hipMemcpyDeviceToHost, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  if (kstep) {
    /* init_kmax (:1031-1045): 2 pi/L with L the distance between the first and last cell centres */
    double dx = 1./dom->lay[dom->depth].n;
    *kstep = 2.*M_PI/((W.N - 1)*dx);
  }
  return GFSHIP_OK;
}

// GfsOutputSpectra of a plane of a 3-D box (realdim == 2, modules/fft.c:1101-1160): the plane normal to
// `normal' (0 x, 1 y, 2 z) at coordinate pos, over the whole box in the other two directions at the finest
// level.  order_array (:800-820) puts the flat direction first and keeps the other two in coordinate order:
// out receives N*(N/2 + 1) complex numbers, index ia*(N/2 + 1) + ib with ia the first and ib the second
// in-plane coordinate (x, y for a z plane; x, z for a y plane; y, z for an x plane); write_spectra prints
// k = kstep times the signed index in those directions and 0 in the normal one.
int gfship_output_spectra_plane (gfship_domain * dom, gfship_field v, int normal, double pos, double * out,
				 double * kstep)
{
  GFSHIP_CHECK (dom && out, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (dom->dim == 3, GFSHIP_EUNSUPPORTED, "the spectrum of a plane is that of a plane of a 3-D box");
  GFSHIP_CHECK (!dom->has_external, GFSHIP_EUNSUPPORTED, "the spectrum of a plane is computed on one box");
  GFSHIP_CHECK (normal >= 0 && normal < 3, GFSHIP_EINVAL, "normal %d out of range", normal);
  Field * F = get_field (dom, v);
  if (!F) return GFSHIP_EINVAL;
  const Layout & L = dom->lay[dom->depth];
  const int N = L.n, nh = N/2 + 1;
  /* gfs_domain_locate: the cell whose extent holds the point */
  int kc = (int) floor ((pos + 0.5)*N) + 1;
  GFSHIP_CHECK (kc >= 1 && kc <= N, GFSHIP_EINVAL, "the plane at %g lies outside the box", pos);
  double * a = nullptr, * sum = nullptr;
  double2 * Fo = nullptr;
  hipfftHandle plan = 0;
  int r = GFSHIP_OK;
  hipError_t e = hipMalloc ((void **) &a, (size_t) N*N*sizeof (double));
  if (e == hipSuccess) e = hipMalloc ((void **) &sum, sizeof (double));
  if (e == hipSuccess) e = hipMalloc ((void **) &Fo, (size_t) N*nh*sizeof (double2));
  if (e == hipSuccess) e = hipMemsetAsync (sum, 0, sizeof (double), dom->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL (spectra_plane_gather_kernel, dim3 ((N*N + 255)/256), dim3 (256), 0, dom->stream, L,
			F->lev[dom->depth], normal, kc, a, sum);
    hipLaunchKernelGGL (spectra_plane_center_kernel, dim3 ((N*N + 255)/256), dim3 (256), 0, dom->stream, N*N, sum, a);
    e = hipGetLastError ();
  }
  if (e == hipSuccess) {
    /* fftw_plan_dft_r2c_3d (1, N, N): a 2-D transform */
    if (hipfftPlan2d (&plan, N, N, HIPFFT_D2Z) != HIPFFT_SUCCESS || hipfftSetStream (plan, dom->stream) != HIPFFT_SUCCESS ||
	hipfftExecD2Z (plan, a, (hipfftDoubleComplex *) Fo) != HIPFFT_SUCCESS) {
      gfship::set_error ("hipFFT error in the transform of a plane");
      r = GFSHIP_EHIP;
    }
  }
  if (e == hipSuccess && r == GFSHIP_OK)
    e = hipMemcpyAsync (out, Fo, (size_t) N*nh*sizeof (double2), hipMemcpyDeviceToHost, dom->stream);
  if (e == hipSuccess && r == GFSHIP_OK)
    e = hipStreamSynchronize (dom->stream);
  if (plan) (void) hipfftDestroy (plan);
  (void) hipFree (a); (void) hipFree (sum); (void) hipFree (Fo);
  if (r) return r;
  GFSHIP_HIP (e);
  if (kstep)
    *kstep = 2.*M_PI/((N - 1)*(1./N));     /* init_kmax: 2 pi/(x1 - x0), first to last cell centre */
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

// variable_turbulent_viscosity_event (modules/turbulence.c:1050-1066): leaf cells only; the velocity
// ghost cells are those of the last BC application
int gfship_turbulent_viscosity (gfship_domain * dom, const gfship_field u[3], double Cs, int model,
				gfship_field out)
{
  GFSHIP_CHECK (dom && u, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (model == 0 || model == 1, GFSHIP_EINVAL, "model: 1 Smagorinsky, 0 sigma");
  const double * up[3] = { nullptr, nullptr, nullptr };
  for (int c = 0; c < dom->dim; c++) {
    Field * F = get_field (dom, u[c]);
    if (!F) return GFSHIP_EINVAL;
    up[c] = F->lev[dom->depth];
  }
  Field * O = get_field (dom, out);
  if (!O) return GFSHIP_EINVAL;
  const Layout & L = dom->lay[dom->depth];
  const size_t ntot = dom->dim == 3 ? (size_t) L.n*L.n*L.n : (size_t) L.n*L.n;
  const dim3 grid ((unsigned) ((ntot + 255)/256));
  if (dom->dim == 3)
    hipLaunchKernelGGL (turbulent_viscosity_kernel<3>, grid, dim3 (256), 0, dom->stream, L, up[0], up[1], up[2],
			Cs, model, O->lev[dom->depth]);
  else
    hipLaunchKernelGGL (turbulent_viscosity_kernel<2>, grid, dim3 (256), 0, dom->stream, L, up[0], up[1], up[1],
			Cs, model, O->lev[dom->depth]);
  GFSHIP_HIP (hipGetLastError ());
  O->zero[dom->depth] = false;
  O->coarse_valid = false;
  return GFSHIP_OK;
}

} // extern "C"
