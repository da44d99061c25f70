// poisson_kernels.hip -- CDNA4 kernels of the multigrid Poisson path (K1-K7 of SURVEY.md 2.5).
//
// All arithmetic is IEEE fp64, compiled with -ffp-contract=off, and written operation by
// operation in the operand order of the reference (src/poisson.c, src/fluid.c) so that the
// exact-mode results are bit-identical to the CPU algorithm.  Unit face weights
// (gfs_poisson_coefficients with alpha = NULL on a uniform single box: every f[d].v == 1.).
#include "gfship_internal.hpp"
#include "relax_skew.hpp"
#include <algorithm>
#include <cstdlib>

namespace gfship {

// ---------------------------------------------------------------------------------------------
// K1: relax / relax2D, src/poisson.c:507-557, with face_weighted_gradient's same-level branch
// (src/fluid.c:858-864: g->a = w; g->b = w*u_nb) for w == 1.
//
// OP == 1 is diffusion_relax, src/poisson.c:1455-1484, with gfs_face_cm_weighted_gradient's
// same-level branch (src/fluid.c:1361-1366: g->a = w; g->b = w*u_nb) for the uniform face
// weight w of the level and h2 = h*h:  a = dia*h*h ; g.a = 1. + g.a/a ; u = (g.b/a + res)/g.a.
// ---------------------------------------------------------------------------------------------
struct W6 { const double * p[6]; };   // the six face weights f[d].v of a level (OP == 2)

template <int DIM, int OP>
__device__ __forceinline__ double relax_value (const double * __restrict__ u, long c, long sy, long sz,
					       double rhs, double dia, unsigned dimension,
					       double omega, double w, double h2, const W6 * wf = nullptr)
{
  if (OP == 2) {
    // relax / relax2D with the face weights of the cell (src/poisson.c:507-557, gfs_face_weighted_
    // gradient's same-level branch src/fluid.c:858-864: g.a = w, g.b = w*u_nb)
    double a = dia, b = 0.;
    { const double g = wf->p[0][c]; a += g; b += g*u[c + 1]; }
    { const double g = wf->p[1][c]; a += g; b += g*u[c - 1]; }
    { const double g = wf->p[2][c]; a += g; b += g*u[c + sy]; }
    { const double g = wf->p[3][c]; a += g; b += g*u[c - sy]; }
    if (DIM == 3) {
      { const double g = wf->p[4][c]; a += g; b += g*u[c + sz]; }
      { const double g = wf->p[5][c]; a += g; b += g*u[c - sz]; }
    }
    if (dimension == 2)
      return a != 0. ? (1. - omega)*u[c] + omega*(b - rhs)/a : 0.;
    return a != 0. ? (b - rhs)/a : 0.;
  }
  if (OP == 1) {
    double ga = 0., gb = 0.;
    ga += w; gb += w*u[c + 1];
    ga += w; gb += w*u[c - 1];
    ga += w; gb += w*u[c + sy];
    ga += w; gb += w*u[c - sy];
    if (DIM == 3) {
      ga += w; gb += w*u[c + sz];
      ga += w; gb += w*u[c - sz];
    }
    double a = dia*h2;
    ga = 1. + ga/a;
    return (gb/a + rhs)/ga;
  }
  double a = dia, b = 0.;
  a += 1.; b += 1.*u[c + 1];
  a += 1.; b += 1.*u[c - 1];
  a += 1.; b += 1.*u[c + sy];
  a += 1.; b += 1.*u[c - sy];
  if (DIM == 3) {
    a += 1.; b += 1.*u[c + sz];
    a += 1.; b += 1.*u[c - sz];
  }
  if (dimension == 2)
    return a != 0. ? (1. - omega)*u[c] + omega*(b - rhs)/a : 0.;
  return a != 0. ? (b - rhs)/a : 0.;
}

// the same for unit weights, dia == 0 and dimension == 3: a = 0. + 1. + ... + 1. = 6. exactly and the
// correctly rounded quotient comes from divide_by_6's reciprocal sequence (relax_skew.hpp) instead of
// the division
__device__ __forceinline__ double relax_value_six (const double * __restrict__ u, long c, long sy, long sz,
						    double rhs)
{
  double b = 0.;
  b += 1.*u[c + 1];
  b += 1.*u[c - 1];
  b += 1.*u[c + sy];
  b += 1.*u[c - sy];
  b += 1.*u[c + sz];
  b += 1.*u[c - sz];
  return divide_by_6 (b - rhs);
}

// Exact-order sweep, one launch per hyperplane I + J + K = plane of the oriented coordinates
// I = i - 1, J = n - j, K = n - k: the reference visits cells in tree pre-order with children
// n = 0..7 at x:+ (bit0), y:- (bit1), z:- (bit2) (src/ftt.c:301-316,837-852); for a 7-point
// stencil the result of the in-place sweep only depends on which of two neighbouring cells
// is visited first, and that orientation is (+x, -y, -z) for every pair, so any topological
// order of it -- here hyperplanes -- is bit-identical (tests/test_oracle_golden_poisson.py
// checks the claim on the oracle).
template <int DIM, int OP>
__global__ void __launch_bounds__(256)
relax_hyperplane_kernel (Layout L, int plane, unsigned dimension, double omega, double w, double h2,
			 double * __restrict__ u, const double * __restrict__ rhs,
			 const double * __restrict__ dia, W6 wf)
{
  int n = L.n;
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  int J, K, I;
  if (DIM == 3) {
    if (t >= n*n) return;
    J = t % n; K = t / n;
    I = plane - J - K;
  }
  else {
    if (t >= n) return;
    J = t; K = 0;
    I = plane - J;
  }
  if (I < 0 || I >= n) return;
  long c = L.idx (I + 1, n - J, DIM == 3 ? n - K : 0);
  u[c] = relax_value<DIM, OP> (u, c, L.sy, L.sz, rhs[c], dia[c], dimension, omega, w, h2, &wf);
}

// ---------------------------------------------------------------------------------------------
// The exact-order sweep of a 2-D level in ONE launch (relax2D, src/poisson.c:532-557, and
// diffusion_relax in 2-D): one workgroup, one thread per row.  Thread J owns the row j = n - J and
// walks it in the direction of the sweep, one step behind thread J - 1: at step t it updates the
// cell I = t - J (i = I + 1) -- the hyperplane I + J = t of the oriented coordinates, all of whose
// cells are independent (launch_relax_exact's order, one launch per hyperplane before: 2 n - 1
// launches per sweep, 4.2 s for the 128^2 run of test/periodic).  What a cell needs from other rows
// travels through LDS, double-buffered by the parity of the step (one barrier per step): the new
// value of the row above (thread J - 1, computed one step earlier) and the old value of the row
// below (thread J + 1 publishes what it holds as its `right' neighbour for its next step).  A
// thread reads its own row ahead of its stores (RX_D steps: a register ring, the step loop unrolled
// by RX_D) and nobody else's, so no global-memory ordering between threads is involved; the ghost
// rows j = n + 1 and j = 0 (written by the BC application before the sweep, not by the sweep) are
// streamed by the threads of the first and the last row.  Same expressions as relax_value<2, OP>.
// ---------------------------------------------------------------------------------------------
#define RX_D 8

template <int OP>
__global__ void __launch_bounds__(1024)
relax_rows2d_kernel (Layout L, double omega, double w, double h2,
		     double * __restrict__ u, const double * __restrict__ rhs,
		     const double * __restrict__ dia, W6 wf)
{
  __shared__ double N[2][1024 + 2], O[2][1024 + 2];
  const int n = L.n;
  const int J = threadIdx.x;
  const bool mine = J < n;
  const int j = mine ? n - J : 1;
  double * const row = u + L.idx (0, j, 0) - 1;           // row[i + 1] = u (i, j): i = -1 is never used
  const double * const rrow = rhs + L.idx (0, j, 0) - 1;
  const double * const drow = dia + L.idx (0, j, 0) - 1;
  // the ghost row beside the first / the last row of the sweep
  const double * const erow = J == 0 ? u + L.idx (0, n + 1, 0) - 1 :
    J == n - 1 ? u + L.idx (0, 0, 0) - 1 : nullptr;
  auto clampi = [n] (int i) { return i < 0 ? 0 : i > n + 1 ? n + 1 : i; };
  // at step t: I = t - J; right = u (I + 2, j) = row[I + 3]; rhs, dia, ghost row at i = I + 1
  double pR[RX_D], pH[RX_D], pD[RX_D], pE[RX_D];
  // OP == 2: the four face weights of the cell (rows of the natural arrays f[d].v)
  double pW[4][RX_D];
  const double * wrow[4];
#pragma unroll
  for (int d = 0; d < 4; d++)
    wrow[d] = OP == 2 ? wf.p[d] + L.idx (0, j, 0) - 1 : nullptr;
#pragma unroll
  for (int q = 0; q < RX_D; q++) {
    const int i = q - J + 1;
    pR[q] = row[clampi (i + 1) + 1];
    pH[q] = rrow[clampi (i) + 1];
    pD[q] = drow[clampi (i) + 1];
    pE[q] = erow ? erow[clampi (i) + 1] : 0.;
    if (OP == 2) {
#pragma unroll
      for (int d = 0; d < 4; d++)
	pW[d][q] = wrow[d][clampi (i) + 1];
    }
  }
  double left = row[0 + 1], cur = row[1 + 1];
  if (mine) {
    // what thread J - 1 reads at step 0 as the old value below its cell I = - (J - 1): only the
    // thread J = 1 is read then (I = 0 of thread 0): its cell i = 1
    O[0][J + 1] = cur;
    N[0][J + 1] = 0.;
  }
  __syncthreads ();
  const int T = (2*n - 1 + RX_D - 1)/RX_D*RX_D;
  for (int t0 = 0; t0 < T; t0 += RX_D) {
#pragma unroll
    for (int q = 0; q < RX_D; q++) {
      const int t = t0 + q, I = t - J;
      const int rd = t & 1, wr = rd ^ 1;
      const double right = pR[q], rh = pH[q], di = pD[q], ex = pE[q];
      double w0 = 0., w1 = 0., w2 = 0., w3 = 0.;
      if (OP == 2) { w0 = pW[0][q]; w1 = pW[1][q]; w2 = pW[2][q]; w3 = pW[3][q]; }
      // the loads of step t + RX_D
      {
	const int i = I + RX_D + 1;
	pR[q] = row[clampi (i + 1) + 1];
	pH[q] = rrow[clampi (i) + 1];
	pD[q] = drow[clampi (i) + 1];
	if (erow) pE[q] = erow[clampi (i) + 1];
	if (OP == 2) {
#pragma unroll
	  for (int d = 0; d < 4; d++)
	    pW[d][q] = wrow[d][clampi (i) + 1];
	}
      }
      double v = 0.;
      const bool active = mine && I >= 0 && I < n;
      if (active) {
	const double top = J == 0 ? ex : N[rd][J];              // new value of (I, J - 1): row j + 1
	const double bottom = J == n - 1 ? ex : O[rd][J + 2];     // old value of (I, J + 1): row j - 1
	if (OP == 1) {
	  double ga = 0., gb = 0.;
	  ga += w; gb += w*right;
	  ga += w; gb += w*left;
	  ga += w; gb += w*top;
	  ga += w; gb += w*bottom;
	  double a = di*h2;
	  ga = 1. + ga/a;
	  v = (gb/a + rh)/ga;
	}
	else if (OP == 2) {
	  // relax2D with the face weights of the cell (src/poisson.c:532-557, src/fluid.c:858-864)
	  double a = di, b = 0.;
	  a += w0; b += w0*right;
	  a += w1; b += w1*left;
	  a += w2; b += w2*top;
	  a += w3; b += w3*bottom;
	  v = a != 0. ? (1. - omega)*cur + omega*(b - rh)/a : 0.;
	}
	else {
	  double a = di, b = 0.;
	  a += 1.; b += 1.*right;
	  a += 1.; b += 1.*left;
	  a += 1.; b += 1.*top;
	  a += 1.; b += 1.*bottom;
	  v = a != 0. ? (1. - omega)*cur + omega*(b - rh)/a : 0.;
	}
	row[I + 1 + 1] = v;
	left = v;
	cur = right;
      }
      if (mine) {
	N[wr][J + 1] = v;
	// the old value thread J - 1 needs below its cell of step t + 1: my `right' of step t + 1
	O[wr][J + 1] = pR[(q + 1) % RX_D];
      }
      __syncthreads ();
    }
  }
}

static int launch_relax_rows2d (gfship_domain * dom, int level, double omega, double * u,
				const double * rhs, const double * dia, int kind, double w, double h2,
				const RelaxOp * op)
{
  const Layout & L = dom->lay[level];
  const int block = L.n <= 64 ? 64 : L.n <= 128 ? 128 : L.n <= 256 ? 256 : L.n <= 512 ? 512 : 1024;
  W6 wf;
  for (int d = 0; d < 6; d++) wf.p[d] = op ? op->wf[d] : nullptr;
  if (kind == 2)
    hipLaunchKernelGGL (relax_rows2d_kernel<2>, dim3 (1), dim3 (block), 0, dom->stream, L, omega, w, h2,
			u, rhs, dia, wf);
  else if (kind)
    hipLaunchKernelGGL (relax_rows2d_kernel<1>, dim3 (1), dim3 (block), 0, dom->stream, L, omega, w, h2,
			u, rhs, dia, wf);
  else
    hipLaunchKernelGGL (relax_rows2d_kernel<0>, dim3 (1), dim3 (block), 0, dom->stream, L, omega, w, h2,
			u, rhs, dia, wf);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int launch_relax_exact (gfship_domain * dom, unsigned dimension, int level, double omega,
			double * u, const double * rhs, const double * dia, const RelaxOp * op)
{
  const int kind = op ? op->kind : 0;
  const double w = op ? op->w : 1., h2 = op ? op->h2 : 1.;
  const Layout & L = dom->lay[level];
  int n = L.n;
  static const bool rows2d = getenv ("GFSHIP_NO_ROWS2D") == nullptr;
  if (dom->dim == 2 && dimension == 2 && n >= 8 && n <= 1024 && rows2d && !dom->force_hyperplane)
    return launch_relax_rows2d (dom, level, omega, u, rhs, dia, kind, w, h2, op);
  int nplanes = dom->dim == 3 ? 3*n - 2 : 2*n - 1;
  int nthreads = dom->dim == 3 ? n*n : n;
  int block = 256;
  int grid = (nthreads + block - 1)/block;
  W6 wf;
  for (int d = 0; d < 6; d++) wf.p[d] = op ? op->wf[d] : nullptr;
  for (int p = 0; p < nplanes; p++) {
#define HP_LAUNCH(D, O) hipLaunchKernelGGL ((relax_hyperplane_kernel<D, O>), dim3 (grid), dim3 (block), \
					    0, dom->stream, L, p, dimension, omega, w, h2, u, rhs, dia, wf)
    if (dom->dim == 3) { if (kind == 2) HP_LAUNCH (3, 2); else if (kind) HP_LAUNCH (3, 1); else HP_LAUNCH (3, 0); }
    else               { if (kind == 2) HP_LAUNCH (2, 2); else if (kind) HP_LAUNCH (2, 1); else HP_LAUNCH (2, 0); }
#undef HP_LAUNCH
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Ghost-cell values, shared by the BC kernel and the LDS relax loop.
// symmetry src/boundary.c:45-51, dirichlet :253-268, neumann :336-347.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double ghost_value (int type, int component, int c, double nb,
					       int homogeneous, double val, double h)
{
  switch (type) {
  case GFSHIP_BC_DIRICHLET:
    return homogeneous ? - nb : 2.*val - nb;
  case GFSHIP_BC_NEUMANN:
    return homogeneous ? nb : nb + val*h;
  default: /* GFSHIP_BC_SYMMETRY */
    return component == c ? - nb : nb;
  }
}

// ---------------------------------------------------------------------------------------------
// Whole relax_loop (src/poisson.c:1070-1089) of a level that fits in LDS, one workgroup:
//   homogeneous BC ; (nrelax - 1) x { sweep ; homogeneous BC } ; sweep
// The level (with ghosts) lives in LDS with the compact (n+2)^DIM layout; rhs and dia are
// read from global memory.  Hyperplanes are separated by workgroup barriers.
// ---------------------------------------------------------------------------------------------
template <int DIM, int OP>
__global__ void __launch_bounds__(1024)
relax_loop_lds_kernel (Layout L, BcDesc bc, unsigned dimension, double omega, double w, double h2,
		       unsigned nrelax, double * __restrict__ u, const double * __restrict__ rhs,
		       const double * __restrict__ dia, W6 wf)
{
  extern __shared__ double s[];
  const int n = L.n, r = n + 2;
  const long ssy = r, ssz = DIM == 3 ? (long) r*r : 0;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ncell = DIM == 3 ? n*n*n : n*n;
  const int nface = DIM == 3 ? n*n : n;

  // zero the tile (edge/corner ghosts are never written), then load the interior
  for (int q = tid; q < (DIM == 3 ? r*r*r : r*r); q += nt)
    s[q] = 0.;
  __syncthreads ();
  for (int q = tid; q < ncell; q += nt) {
    int i = q % n + 1, j = (q / n) % n + 1, k = DIM == 3 ? q / (n*n) + 1 : 0;
    s[i + ssy*j + ssz*k] = u[L.idx (i, j, k)];
  }
  __syncthreads ();

  for (unsigned sweep = 0; sweep < nrelax; sweep++) {
    // homogeneous BC on the ghost layer (every side reads interior cells only)
    for (int q = tid; q < 2*DIM*nface; q += nt) {
      int d = q / nface, f = q % nface;
      int c = d/2;
      int t1 = f % n + 1, t2 = DIM == 3 ? f / n + 1 : 0;
      int ijk[3] = { 0, 0, 0 };
      int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
      ijk[c] = (d & 1) ? 1 : n;
      ijk[ta] = t1;
      if (DIM == 3) ijk[tb] = t2;
      long o = c == 0 ? 1 : c == 1 ? ssy : ssz;
      if (d & 1) o = - o;
      long nb = ijk[0] + ssy*ijk[1] + ssz*ijk[2];
      double v;
      if (bc.side[d] == GFSHIP_SIDE_PERIODIC)
	v = s[nb - (long) (n - 1)*o];
      else if (bc.side[d] == GFSHIP_SIDE_EXTERNAL) {
	/* GfsBoundaryMpi: the ghost holds what the neighbour box sent before this launch (single
	   sweep per launch on such domains) */
	int gi[3] = { ijk[0], ijk[1], ijk[2] };
	gi[c] += (d & 1) ? -1 : 1;
	v = u[L.idx (gi[0], gi[1], gi[2])];
      }
      else
	v = ghost_value (bc.type[d], bc.component, c, s[nb], 1, 0., 0.);
      s[nb + o] = v;
    }
    __syncthreads ();
    // exact-order sweep by hyperplanes
    const int nplanes = DIM == 3 ? 3*n - 2 : 2*n - 1;
    for (int plane = 0; plane < nplanes; plane++) {
      for (int t = tid; t < nface; t += nt) {
	int J = t % n, K = DIM == 3 ? t / n : 0;
	int I = plane - J - K;
	if (I >= 0 && I < n) {
	  int i = I + 1, j = n - J, k = DIM == 3 ? n - K : 0;
	  long c = i + ssy*j + ssz*k;
	  long g = L.idx (i, j, k);
	  if (OP == 2) {
	    // the face weights of the cell live in the natural arrays (index g), u in LDS (index c)
	    double a = dia[g], b = 0.;
	    { const double q = wf.p[0][g]; a += q; b += q*s[c + 1]; }
	    { const double q = wf.p[1][g]; a += q; b += q*s[c - 1]; }
	    { const double q = wf.p[2][g]; a += q; b += q*s[c + ssy]; }
	    { const double q = wf.p[3][g]; a += q; b += q*s[c - ssy]; }
	    if (DIM == 3) {
	      { const double q = wf.p[4][g]; a += q; b += q*s[c + ssz]; }
	      { const double q = wf.p[5][g]; a += q; b += q*s[c - ssz]; }
	    }
	    s[c] = dimension == 2 ? (a != 0. ? (1. - omega)*s[c] + omega*(b - rhs[g])/a : 0.) :
	      (a != 0. ? (b - rhs[g])/a : 0.);
	  }
	  else
	    s[c] = relax_value<DIM, OP> (s, c, ssy, ssz, rhs[g], dia[g], dimension, omega, w, h2);
	}
      }
      __syncthreads ();
    }
  }
  // store interior and ghost layer (the reference's ghost cells keep the values of the last
  // BC application, which get_from_above then reads: src/poisson.c:1160-1167)
  const int nall = DIM == 3 ? r*r*r : r*r;
  for (int q = tid; q < nall; q += nt) {
    int i = q % r, j = (q / r) % r, k = DIM == 3 ? q / (r*r) : 0;
    u[L.idx (i, j, k)] = s[q];
  }
}

static bool lds_fits (const gfship_domain * dom, int level, size_t * bytes)
{
  const Layout & L = dom->lay[level];
  size_t r = L.n + 2;
  size_t b = (dom->dim == 3 ? r*r*r : r*r)*sizeof (double);
  *bytes = b;
  return b <= 64*1024; /* stay within the default dynamic-LDS limit */
}

int launch_relax_loop_small (gfship_domain * dom, unsigned dimension, int level, double omega,
			     Field * dp, Field * ubc, const double * rhs, const double * dia,
			     unsigned nrelax, bool * done, const RelaxOp * op)
{
  const int kind = op ? op->kind : 0;
  const double w = op ? op->w : 1., h2 = op ? op->h2 : 1.;
  size_t bytes;
  *done = false;
  for (int d = 0; d < 2*dom->dim; d++)
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL && nrelax > 1)
      return GFSHIP_OK; /* halos need the exchange between sweeps: one sweep per launch only */
  if (!lds_fits (dom, level, &bytes))
    return GFSHIP_OK;
  const Layout & L = dom->lay[level];
  BcDesc bc;
  for (int d = 0; d < 6; d++) {
    bc.side[d] = dom->side[d];
    bc.type[d] = ubc->bc[d];
    bc.val[d] = nullptr;
  }
  bc.component = ubc->component;
  bc.homogeneous = 1;
  int nface = dom->dim == 3 ? L.n*L.n : L.n;
  int block = nface <= 64 ? 64 : nface <= 256 ? 256 : 1024;
  W6 wf;
  for (int d = 0; d < 6; d++) wf.p[d] = op ? op->wf[d] : nullptr;
#define LDS_LAUNCH(D, O) hipLaunchKernelGGL ((relax_loop_lds_kernel<D, O>), dim3 (1), dim3 (block), bytes, \
					     dom->stream, L, bc, dimension, omega, w, h2, nrelax, \
					     dp->lev[level], rhs, dia, wf)
  if (dom->dim == 3) { if (kind == 2) LDS_LAUNCH (3, 2); else if (kind) LDS_LAUNCH (3, 1); else LDS_LAUNCH (3, 0); }
  else               { if (kind == 2) LDS_LAUNCH (2, 2); else if (kind) LDS_LAUNCH (2, 1); else LDS_LAUNCH (2, 0); }
#undef LDS_LAUNCH
  GFSHIP_HIP (hipGetLastError ());
  *done = true;
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// The coarse end of a V-cycle in one launch (one workgroup, everything in LDS): the levels
// lmin .. ltop whose relax loops would otherwise be one launch each plus their restrictions,
// prolongations, fills and BC kernels (src/poisson.c:1131-1168):
//   res[l] = get_from_below (res[l+1])            l = ltop .. lmin   (res[ltop+1] from global)
//   dp[lmin] = 0 ; relax_loop (dp, lmin)
//   dp[l] = get_from_above (dp[l-1]) ; relax_loop (dp, l)             l = lmin+1 .. ltop
// Results: dp[ltop] with its ghost layer (what the prolongation to ltop+1 reads) and the
// restricted residuals, written to global memory in the natural layout.  Same arithmetic and
// same order per cell as the per-level kernels (relax_loop_lds_kernel, restrict_kernel,
// prolongate_kernel).
// ---------------------------------------------------------------------------------------------
struct CoarseCycleArgs {
  Layout lay[GFSHIP_MAXLEVEL + 1];
  int lmin, ltop;
  unsigned nrelax[GFSHIP_MAXLEVEL + 1];
  double * dp[GFSHIP_MAXLEVEL + 1];          // natural, all levels lmin..ltop are written
  double * res[GFSHIP_MAXLEVEL + 2];         // natural; res[ltop + 1] is the input
  const double * dia[GFSHIP_MAXLEVEL + 1];   // nullptr: dia == 0 on that level
  BcDesc bc;
  unsigned dimension;
  double omega;
};

template <int DIM>
__global__ void __launch_bounds__(1024)
coarse_cycle_kernel (CoarseCycleArgs A)
{
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
#ifdef GFSHIP_LAB_CLOCK
  unsigned long long stamp[40]; int nstamp = 0;
#define LAB_STAMP() do { if (nstamp < 40) stamp[nstamp++] = wall_clock64 (); } while (0)
#else
#define LAB_STAMP() do { } while (0)
#endif
  LAB_STAMP ();
  // LDS map: per level a dp array and a res array of (n+2)^DIM doubles
  // offsets, not pointers kept in an array: the accesses stay LDS instructions (an array of
  // pointers indexed by the level lives in scratch and turns every access into a flat one)
  auto lds_off = [&] (int l, int which) -> unsigned {
    unsigned o = 0;
    for (int q = A.lmin; q < l; q++) {
      unsigned r = A.lay[q].n + 2;
      o += 2*(DIM == 3 ? r*r*r : r*r);
    }
    if (which) {
      unsigned r = A.lay[l].n + 2;
      o += DIM == 3 ? r*r*r : r*r;
    }
    return o;
  };
  {
    const unsigned o = lds_off (A.ltop + 1, 0);
    for (size_t q = tid; q < o; q += nt)
      lds[q] = 0.;
  }
  __syncthreads ();
  LAB_STAMP ();
  // ---- restrictions, finest first ----
  for (int l = A.ltop; l >= A.lmin; l--) {
    const Layout & Lc = A.lay[l], & Lf = A.lay[l + 1];
    const int n = Lc.n, r = n + 2, rf = 2*n + 2;
    const int ncell = DIM == 3 ? n*n*n : n*n;
    const bool top = l == A.ltop;
    const double * gf = A.res[l + 1];
    const double * sf = lds + (top ? 0 : lds_off (l + 1, 1));
    double * sc = lds + lds_off (l, 1);
    for (int q = tid; q < ncell; q += nt) {
      int i = q % n + 1, j = (q / n) % n + 1, k = DIM == 3 ? q / (n*n) + 1 : 0;
      double val = 0.;
#pragma unroll
      for (int id = 0; id < (1 << DIM); id++) {
	int ci = 2*i - 1 + (id & 1);
	int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
	int ck = DIM == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
	if (top)
	  val += gf[Lf.idx (ci, cj, ck)];
	else
	  val += sf[ci + rf*(cj + (DIM == 3 ? rf*ck : 0))];
      }
      double v = A.dimension == 2 ? val : val/2.;
      sc[i + r*(j + (DIM == 3 ? r*k : 0))] = v;
      A.res[l][Lc.idx (i, j, k)] = v;
    }
    __syncthreads ();
    LAB_STAMP ();
  }
  // ---- relax loops from the coarsest level up ----
  for (int l = A.lmin; l <= A.ltop; l++) {
    const Layout & L = A.lay[l];
    const int n = L.n, r = n + 2;
    const long ssy = r, ssz = DIM == 3 ? (long) r*r : 0;
    const int ncell = DIM == 3 ? n*n*n : n*n;
    const int nface = DIM == 3 ? n*n : n;
    double * s = (lds + lds_off (l, 0));
    if (l > A.lmin) {
      // get_from_above, src/poisson.c:1005-1042 (prolongate_kernel), one thread per fine cell
      const int nc = n/2, rc = nc + 2;
      const long cy = rc, cz = DIM == 3 ? (long) rc*rc : 0;
      const double * vc = (lds + lds_off (l - 1, 0));
      for (int q = tid; q < ncell; q += nt) {
	int i = q % n + 1, j = (q / n) % n + 1, k = DIM == 3 ? q / (n*n) + 1 : 0;
	int pi = (i + 1)/2, pj = (j + 1)/2, pk = DIM == 3 ? (k + 1)/2 : 0;
	long p = pi + cy*pj + cz*pk;
	double pv = vc[p];
	double h[3];
	const long off[3] = { 1, cy, cz };
#pragma unroll
	for (int cc = 0; cc < DIM; cc++) {
	  double g1 = vc[p + off[cc]] - 1.*pv;
	  double g2 = vc[p - off[cc]] - 1.*pv;
	  h[cc] = (g1 - g2)/2.;
	}
	double rel[3] = { ((i & 1) ? -1. : 1.)/4., ((j & 1) ? -1. : 1.)/4., ((k & 1) ? -1. : 1.)/4. };
	double val = pv;
#pragma unroll
	for (int cc = 0; cc < DIM; cc++)
	  val += rel[cc]*h[cc];
	s[i + ssy*j + ssz*k] = val;
      }
      __syncthreads ();
    }
    LAB_STAMP ();
    const double * rhs = (lds + lds_off (l, 1));
    const double * dia = A.dia[l];
    const int S = (int) A.nrelax[l];
    if (S > 0) {
      // the BC before the first sweep, from the values the level starts with
      for (int q = tid; q < 2*DIM*nface; q += nt) {
	int d = q / nface, f = q % nface;
	int c = d/2;
	int t1 = f % n + 1, t2 = DIM == 3 ? f / n + 1 : 0;
	int ijk[3] = { 0, 0, 0 };
	int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
	ijk[c] = (d & 1) ? 1 : n;
	ijk[ta] = t1;
	if (DIM == 3) ijk[tb] = t2;
	long o = c == 0 ? 1 : c == 1 ? ssy : ssz;
	if (d & 1) o = - o;
	long nb = ijk[0] + ssy*ijk[1] + ssz*ijk[2];
	double v;
	if (A.bc.side[d] == GFSHIP_SIDE_PERIODIC)
	  v = s[nb - (long) (n - 1)*o];
	else
	  v = ghost_value (A.bc.type[d], A.bc.component, c, s[nb], 1, 0., 0.);
	s[nb + o] = v;
      }
      __syncthreads ();
      // The sweeps of the loop follow each other n planes apart instead of one after the other:
      // sweep s is on plane q - s*n at iteration q.  A cell of sweep s+1 on plane p needs its
      // later neighbours (plane p + 1) from sweep s -- two planes would do -- and the ghosts of its
      // lines as the BC between the two sweeps leaves them: the image of the line's last cell,
      // which sweep s reaches n - 1 planes after the line's first.  The thread that computes the
      // last cell of a line refreshes both ghosts of the line then (sweep s has read them, sweep
      // s+1 reaches the line one iteration later at the earliest); the last sweep refreshes
      // none, so the ghosts stay those of the last BC application.  Same values in the same
      // order as sweep after sweep, in half the barriers.
      const int nplanes = DIM == 3 ? 3*n - 2 : 2*n - 1;
      const int nconc = (nplanes + n - 1)/n;
      const int niter = nplanes + (S - 1)*n;
      const bool per0 = A.bc.side[0] == GFSHIP_SIDE_PERIODIC, per1 = A.bc.side[1] == GFSHIP_SIDE_PERIODIC;
      const bool per2 = A.bc.side[2] == GFSHIP_SIDE_PERIODIC, per3 = A.bc.side[3] == GFSHIP_SIDE_PERIODIC;
      const bool per4 = DIM == 3 && A.bc.side[4] == GFSHIP_SIDE_PERIODIC;
      const bool per5 = DIM == 3 && A.bc.side[5] == GFSHIP_SIDE_PERIODIC;
      const int lg = __ffs (n) - 1, lgf = DIM == 3 ? 2*lg : lg; // n is a power of two
      auto cell = [&] (int sw, int I, int J, int K) {
	int i = I + 1, j = n - J, k = DIM == 3 ? n - K : 0;
	long c = i + ssy*j + ssz*k;
	double v;
	if (DIM == 3 && !dia && A.dimension == 3)
	  v = relax_value_six (s, c, ssy, ssz, rhs[c]);
	else {
	  double dv = dia ? dia[L.idx (i, j, k)] : 0.;
	  v = relax_value<DIM, 0> (s, c, ssy, ssz, rhs[c], dv, A.dimension, A.omega, 1., 1.);
	}
	s[c] = v;
	if (sw < S - 1) {
	  if (i == n) {
	    long f = 1 + ssy*j + ssz*k;
	    double first = s[f];
	    s[f - 1] = per1 ? v : ghost_value (A.bc.type[1], A.bc.component, 0, first, 1, 0., 0.);
	    s[c + 1] = per0 ? first : ghost_value (A.bc.type[0], A.bc.component, 0, v, 1, 0., 0.);
	  }
	  if (j == 1) {
	    long f = i + ssy*n + ssz*k;
	    double first = s[f];
	    s[f + ssy] = per2 ? v : ghost_value (A.bc.type[2], A.bc.component, 1, first, 1, 0., 0.);
	    s[c - ssy] = per3 ? first : ghost_value (A.bc.type[3], A.bc.component, 1, v, 1, 0., 0.);
	  }
	  if (DIM == 3 && k == 1) {
	    long f = i + ssy*j + ssz*n;
	    double first = s[f];
	    s[f + ssz] = per4 ? v : ghost_value (A.bc.type[4], A.bc.component, 2, first, 1, 0., 0.);
	    s[c - ssz] = per5 ? first : ghost_value (A.bc.type[5], A.bc.component, 2, v, 1, 0., 0.);
	  }
	}
      };
      if (DIM == 3 && !dia && A.dimension == 3 && nconc*nface <= nt) {
	// the usual case, one (sweep slot, line of x) per thread for the whole loop: what does not
	// change from plane to plane is computed once, an iteration is the stencil and a barrier
	const int g = tid >> lgf, t = tid & (nface - 1);
	const int J = t & (n - 1), K = t >> lg;
	const int j = n - J, k = n - K;
	double * srow = s + (ssy*j + ssz*k);
	const double * rrow = rhs + (ssy*j + ssz*k);
	double * sj = s + (ssy*n + ssz*k), * sk = s + (ssy*j + ssz*n); // first lines along y and z
	const bool mine = g < nconc, jl = j == 1, kl = k == 1;
	for (int q = 0; q < niter; q++) {
	  const int sw = min (S - 1, q >> lg) - g;
	  const int I = q - (sw << lg) - J - K;
	  if (mine && sw >= 0 && (unsigned) I < (unsigned) n) {
	    const int i = I + 1;
	    double b = 0.;
	    b += 1.*srow[i + 1];
	    b += 1.*srow[i - 1];
	    b += 1.*srow[i + ssy];
	    b += 1.*srow[i - ssy];
	    b += 1.*srow[i + ssz];
	    b += 1.*srow[i - ssz];
	    const double v = divide_by_6 (b - rrow[i]);
	    srow[i] = v;
	    if (sw < S - 1 && (i == n || jl || kl)) {
	      if (i == n) {
		double first = srow[1];
		srow[0] = per1 ? v : ghost_value (A.bc.type[1], A.bc.component, 0, first, 1, 0., 0.);
		srow[n + 1] = per0 ? first : ghost_value (A.bc.type[0], A.bc.component, 0, v, 1, 0., 0.);
	      }
	      if (jl) {
		double first = sj[i];
		sj[i + ssy] = per2 ? v : ghost_value (A.bc.type[2], A.bc.component, 1, first, 1, 0., 0.);
		srow[i - ssy] = per3 ? first : ghost_value (A.bc.type[3], A.bc.component, 1, v, 1, 0., 0.);
	      }
	      if (kl) {
		double first = sk[i];
		sk[i + ssz] = per4 ? v : ghost_value (A.bc.type[4], A.bc.component, 2, first, 1, 0., 0.);
		srow[i - ssz] = per5 ? first : ghost_value (A.bc.type[5], A.bc.component, 2, v, 1, 0., 0.);
	      }
	    }
	  }
	  __syncthreads ();
	}
      }
      else if (nconc*nface <= nt) {
	const int g = tid >> lgf, t = tid & (nface - 1);
	const int J = t & (n - 1), K = DIM == 3 ? t >> lg : 0;
	const bool mine = g < nconc;
	for (int q = 0; q < niter; q++) {
	  const int sw = min (S - 1, q >> lg) - g;
	  const int I = q - (sw << lg) - J - K;
	  if (mine && sw >= 0 && I >= 0 && I < n)
	    cell (sw, I, J, K);
	  __syncthreads ();
	}
      }
      else
	for (int q = 0; q < niter; q++) {
	  const int shi = min (S - 1, q >> lg);
	  for (int w = tid; w < nconc*nface; w += nt) {
	    const int sw = shi - (w >> lgf), t = w & (nface - 1);
	    const int J = t & (n - 1), K = DIM == 3 ? t >> lg : 0;
	    const int I = q - (sw << lg) - J - K;
	    if (sw >= 0 && I >= 0 && I < n)
	      cell (sw, I, J, K);
	  }
	  __syncthreads ();
	}
    }
    LAB_STAMP ();
    // natural copy with the ghost layer of the last BC application
    const int nall = DIM == 3 ? r*r*r : r*r;
    for (int q = tid; q < nall; q += nt) {
      int i = q % r, j = (q / r) % r, k = DIM == 3 ? q / (r*r) : 0;
      A.dp[l][L.idx (i, j, k)] = s[q];
    }
    __syncthreads ();
    LAB_STAMP ();
  }
#ifdef GFSHIP_LAB_CLOCK
  if (tid == 0) {
    for (int q = 1; q < nstamp; q++)
      printf ("coarse stamp %d: %.2f us\n", q, (double) (stamp[q] - stamp[q - 1])/100.);
  }
#endif
#undef LAB_STAMP
}

// highest level of the coarse end that fits (0 .. ltop in LDS); -1 when it does not apply
int coarse_cycle_top (gfship_domain * dom, int minlevel)
{
  for (int d = 0; d < 2*dom->dim; d++)
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL)
      return -1;      /* halos need the exchange between sweeps */
  if (dom->relax_mode != GFSHIP_RELAX_EXACT || dom->force_hyperplane || dom->no_fused_loop)
    return -1;
  int top = -1;
  size_t bytes = 0;
  for (int l = minlevel; l < dom->depth; l++) {     /* the leaf level is never part of it */
    if (dom->dim == 3 && skew_supported (dom, l)) break;
    size_t r = dom->lay[l].n + 2;
    bytes += 2*(dom->dim == 3 ? r*r*r : r*r)*sizeof (double);
    if (bytes > 150*1024) break;
    top = l;
  }
  return top > minlevel ? top : -1;   /* a single level: the plain LDS loop does it */
}

static int coarse_threads ()
{
  static int n = 0;
  if (!n) {
    const char * e = getenv ("GFSHIP_COARSE_THREADS");
    n = e ? atoi (e) : 1024;
    if (n < 64 || n > 1024 || (n & 63)) n = 1024;
  }
  return n;
}

int launch_coarse_cycle (gfship_domain * dom, unsigned dimension, double omega, int lmin, int ltop,
			 const unsigned * nrelax, Field * dp, Field * ubc, Field * res, Field * dia)
{
  CoarseCycleArgs A;
  size_t bytes = 0;
  for (int l = 0; l <= GFSHIP_MAXLEVEL; l++) {
    A.lay[l] = dom->lay[l <= dom->depth ? l : dom->depth];
    A.nrelax[l] = 0; A.dp[l] = nullptr; A.res[l] = nullptr; A.dia[l] = nullptr;
  }
  A.res[GFSHIP_MAXLEVEL + 1] = nullptr;
  for (int l = lmin; l <= ltop; l++) {
    size_t r = dom->lay[l].n + 2;
    bytes += 2*(dom->dim == 3 ? r*r*r : r*r)*sizeof (double);
    A.nrelax[l] = nrelax[l];
    A.dp[l] = dp->lev[l];
    A.res[l] = res->lev[l];
    A.dia[l] = dia->zero[l] ? nullptr : dia->lev[l];
    dp->zero[l] = false;
    res->zero[l] = false;
  }
  A.res[ltop + 1] = res->lev[ltop + 1];
  A.lmin = lmin; A.ltop = ltop;
  for (int d = 0; d < 6; d++) {
    A.bc.side[d] = dom->side[d];
    A.bc.type[d] = ubc->bc[d];
    A.bc.val[d] = nullptr;
  }
  A.bc.component = ubc->component;
  A.bc.homogeneous = 1;
  A.dimension = dimension;
  A.omega = omega;
  if (dom->dim == 3) {
    if (!dom->coarse_attr_set) {
      GFSHIP_HIP (hipFuncSetAttribute ((const void *) coarse_cycle_kernel<3>,
				       hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
      GFSHIP_HIP (hipFuncSetAttribute ((const void *) coarse_cycle_kernel<2>,
				       hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
      dom->coarse_attr_set = true;
    }
    hipLaunchKernelGGL (coarse_cycle_kernel<3>, dim3 (1), dim3 (coarse_threads ()), bytes, dom->stream, A);
  }
  else {
    if (!dom->coarse_attr_set) {
      GFSHIP_HIP (hipFuncSetAttribute ((const void *) coarse_cycle_kernel<2>,
				       hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
      dom->coarse_attr_set = true;
    }
    hipLaunchKernelGGL (coarse_cycle_kernel<2>, dim3 (1), dim3 (coarse_threads ()), bytes, dom->stream, A);
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// The coarse end of a V-cycle on a lattice of boxes (GfsBoundaryMpi sides): what costs the
// reference -- and the per-level path here -- one halo exchange per sweep and level (20 exchanges
// of a few hundred bytes on the levels up to 16^3, each a pack kernel, a send / receive pair and an
// unpack kernel) is computed for EVERY box of the lattice on EVERY rank after one all-gather of the
// residuals of the level ltop: workgroup b of the launch is box b, with its levels in LDS exactly
// as coarse_cycle_kernel holds them; where the reference applies the BC (before every sweep) the
// workgroups write the layers along their MPI sides to device memory, meet at a barrier and read
// the layers of their neighbours into their ghost cells -- gfs_boundary_send / receive of
// src/mpi_boundary.c:89-222 between workgroups.  Every rank performs the same operations in the
// same order as the owners of the boxes would, so the values of its own box are the ones a run
// with one exchange per sweep produces, bit for bit (tests/test_gpu_multibox.py).
// The B <= 16 workgroups of 1024 threads are all resident (nothing else runs on the stream's
// device at that point of the cycle), which the barrier needs.
// ---------------------------------------------------------------------------------------------
#define LAT_MAXBOXES 16

struct LatticeCycleArgs {
  Layout lay[GFSHIP_MAXLEVEL + 1];
  int lmin, ltop;
  unsigned nrelax[GFSHIP_MAXLEVEL + 1];
  double * dp[GFSHIP_MAXLEVEL + 1];          // natural arrays of the own box, lmin..ltop are written
  double * res[GFSHIP_MAXLEVEL + 1];         // natural arrays of the own box, lmin..ltop-1 are written
  const double * gres;                       // res[ltop] of every box (natural layout), in rank order
  unsigned long long * xch;                  // [2][B][6][nface (ltop)] layers between the boxes
  unsigned * bar;                            // barrier counter, zero at launch
  unsigned * err;                            // host-mapped word: set when a barrier wait gives up
  BcDesc bc;
  unsigned dimension;
  double omega;
  int rank, nboxes, b[3];
};

typedef __attribute__((address_space(1))) unsigned long long lat_gu64;

// The layers travel as agent-scope atomics (coherent per location across the XCDs, no cache to flush):
// a thread only has to wait until its own stores have been acknowledged (s_waitcnt vmcnt (0)) before
// the workgroup's barrier, then one thread announces the workgroup and waits for the others.  (With
// __threadfence () on both sides -- a write-back and an invalidate of the whole L2 each -- a barrier
// cost 8 us.)
// The wait is bounded (2^22 polls, tens of milliseconds): should the workgroups not all be resident
// after all (lattice_cycle_top checks the occupancy, but the device may be shared) the kernel sets
// *err, stops waiting at every later barrier too and ends -- the solve is reported as failed
// (lattice_check_error) instead of hanging the device.
__device__ __forceinline__ void lattice_barrier (unsigned * bar, unsigned & target, int nboxes,
						 unsigned * err)
{
  asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads ();
  if (threadIdx.x == 0) {
    target += (unsigned) nboxes;
    __hip_atomic_fetch_add (bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load (bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep (1);
      if (++spins > (1u << 22) ||
	  __hip_atomic_load (err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
	__hip_atomic_store (err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	break;
      }
    }
  }
  __syncthreads ();
}

template <int DIM>
__global__ void __launch_bounds__(1024)
lattice_cycle_kernel (LatticeCycleArgs A)
{
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int box = blockIdx.x;
  const bool own = box == A.rank;
  // the boxes across the six sides of this one
  int nbox[6];
  {
    const int bx = A.b[0], by = A.b[1], bz = A.b[2];
    const int cx = box % bx, cy = (box/bx) % by, cz = box/(bx*by);
    for (int d = 0; d < 6; d++) {
      int cc[3] = { cx, cy, cz };
      cc[d/2] += (d & 1) ? -1 : 1;
      cc[0] = (cc[0] + bx) % bx; cc[1] = (cc[1] + by) % by; cc[2] = (cc[2] + bz) % bz;
      nbox[d] = cc[0] + bx*(cc[1] + by*cc[2]);
    }
  }
  unsigned target = 0, parity = 0;
  const size_t xface = DIM == 3 ? (size_t) A.lay[A.ltop].n*A.lay[A.ltop].n : (size_t) A.lay[A.ltop].n;
  // offsets, not pointers kept in an array: the accesses stay LDS instructions (an array of
  // pointers indexed by the level lives in scratch and turns every access into a flat one)
  auto lds_off = [&] (int l, int which) -> unsigned {
    unsigned o = 0;
    for (int q = A.lmin; q < l; q++) {
      unsigned r = A.lay[q].n + 2;
      o += 2*(DIM == 3 ? r*r*r : r*r);
    }
    if (which) {
      unsigned r = A.lay[l].n + 2;
      o += DIM == 3 ? r*r*r : r*r;
    }
    return o;
  };
  {
    const unsigned o = lds_off (A.ltop + 1, 0);
    for (size_t q = tid; q < o; q += nt)
      lds[q] = 0.;
  }
  __syncthreads ();
  // ---- the residual of the level ltop of this box, then the restrictions (restrict_kernel) ----
  {
    const Layout & L = A.lay[A.ltop];
    const int n = L.n, r = n + 2;
    const int ncell = DIM == 3 ? n*n*n : n*n;
    const double * g = A.gres + (size_t) box*L.total;
    double * st = lds + lds_off (A.ltop, 1);
    for (int q = tid; q < ncell; q += nt) {
      int i = q % n + 1, j = (q / n) % n + 1, k = DIM == 3 ? q / (n*n) + 1 : 0;
      st[i + r*(j + (DIM == 3 ? r*k : 0))] = g[L.idx (i, j, k)];
    }
    __syncthreads ();
  }
  for (int l = A.ltop - 1; l >= A.lmin; l--) {
    const Layout & Lc = A.lay[l];
    const int n = Lc.n, r = n + 2, rf = 2*n + 2;
    const int ncell = DIM == 3 ? n*n*n : n*n;
    const double * sf = lds + lds_off (l + 1, 1);
    double * sc = lds + lds_off (l, 1);
    for (int q = tid; q < ncell; q += nt) {
      int i = q % n + 1, j = (q / n) % n + 1, k = DIM == 3 ? q / (n*n) + 1 : 0;
      double val = 0.;
#pragma unroll
      for (int id = 0; id < (1 << DIM); id++) {
	int ci = 2*i - 1 + (id & 1);
	int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
	int ck = DIM == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
	val += sf[ci + rf*(cj + (DIM == 3 ? rf*ck : 0))];
      }
      double v = A.dimension == 2 ? val : val/2.;
      sc[i + r*(j + (DIM == 3 ? r*k : 0))] = v;
      if (own) A.res[l][Lc.idx (i, j, k)] = v;
    }
    __syncthreads ();
  }
  // ---- relax loops from the coarsest level up ----
  for (int l = A.lmin; l <= A.ltop; l++) {
    const Layout & L = A.lay[l];
    const int n = L.n, r = n + 2;
    const long ssy = r, ssz = DIM == 3 ? (long) r*r : 0;
    const int ncell = DIM == 3 ? n*n*n : n*n;
    const int nface = DIM == 3 ? n*n : n;
    double * s = (lds + lds_off (l, 0));
    if (l > A.lmin) {
      // get_from_above, src/poisson.c:1005-1042 (prolongate_kernel), one thread per fine cell
      const int nc = n/2, rc = nc + 2;
      const long cy = rc, cz = DIM == 3 ? (long) rc*rc : 0;
      const double * vc = (lds + lds_off (l - 1, 0));
      for (int q = tid; q < ncell; q += nt) {
	int i = q % n + 1, j = (q / n) % n + 1, k = DIM == 3 ? q / (n*n) + 1 : 0;
	int pi = (i + 1)/2, pj = (j + 1)/2, pk = DIM == 3 ? (k + 1)/2 : 0;
	long p = pi + cy*pj + cz*pk;
	double pv = vc[p];
	double h[3];
	const long off[3] = { 1, cy, cz };
#pragma unroll
	for (int cc = 0; cc < DIM; cc++) {
	  double g1 = vc[p + off[cc]] - 1.*pv;
	  double g2 = vc[p - off[cc]] - 1.*pv;
	  h[cc] = (g1 - g2)/2.;
	}
	double rel[3] = { ((i & 1) ? -1. : 1.)/4., ((j & 1) ? -1. : 1.)/4., ((k & 1) ? -1. : 1.)/4. };
	double val = pv;
#pragma unroll
	for (int cc = 0; cc < DIM; cc++)
	  val += rel[cc]*h[cc];
	s[i + ssy*j + ssz*k] = val;
      }
      __syncthreads ();
    }
    const double * rhs = (lds + lds_off (l, 1));
    for (unsigned sweep = 0; sweep < A.nrelax[l]; sweep++) {
      // BC application: the layers along the MPI sides go out ...
      unsigned long long * const out = A.xch + ((size_t) parity*A.nboxes + box)*6*xface;
      for (int q = tid; q < 2*DIM*nface; q += nt) {
	int d = q / nface, f = q % nface;
	if (A.bc.side[d] != GFSHIP_SIDE_EXTERNAL) continue;
	int c = d/2;
	int t1 = f % n + 1, t2 = DIM == 3 ? f / n + 1 : 0;
	int ijk[3] = { 0, 0, 0 };
	int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
	ijk[c] = (d & 1) ? 1 : n;
	ijk[ta] = t1;
	if (DIM == 3) ijk[tb] = t2;
	__hip_atomic_store ((lat_gu64 *) (out + d*xface + f),
			    (unsigned long long) __double_as_longlong (s[ijk[0] + ssy*ijk[1] + ssz*ijk[2]]),
			    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      lattice_barrier (A.bar, target, A.nboxes, A.err);
      // ... and the ghost cells are filled: local sides as in bc_kernel, MPI sides from the layer the
      // box across the side has just written (its side d ^ 1, same position on the face)
      for (int q = tid; q < 2*DIM*nface; q += nt) {
	int d = q / nface, f = q % nface;
	int c = d/2;
	int t1 = f % n + 1, t2 = DIM == 3 ? f / n + 1 : 0;
	int ijk[3] = { 0, 0, 0 };
	int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
	ijk[c] = (d & 1) ? 1 : n;
	ijk[ta] = t1;
	if (DIM == 3) ijk[tb] = t2;
	long o = c == 0 ? 1 : c == 1 ? ssy : ssz;
	if (d & 1) o = - o;
	long nb = ijk[0] + ssy*ijk[1] + ssz*ijk[2];
	double v;
	if (A.bc.side[d] == GFSHIP_SIDE_PERIODIC)
	  v = s[nb - (long) (n - 1)*o];
	else if (A.bc.side[d] == GFSHIP_SIDE_EXTERNAL) {
	  const unsigned long long * in = A.xch + ((size_t) parity*A.nboxes + nbox[d])*6*xface;
	  v = __longlong_as_double ((long long)
				    __hip_atomic_load ((lat_gu64 *) (in + (d ^ 1)*xface + f), __ATOMIC_RELAXED,
						       __HIP_MEMORY_SCOPE_AGENT));
	}
	else
	  v = ghost_value (A.bc.type[d], A.bc.component, c, s[nb], 1, 0., 0.);
	s[nb + o] = v;
      }
      parity ^= 1;
      __syncthreads ();
      const int nplanes = DIM == 3 ? 3*n - 2 : 2*n - 1;
      if (DIM == 3 && A.dimension == 3 && nface <= nt) {
	// one line of x per thread, the addresses of the line computed once
	const int lg = __ffs (n) - 1; // n is a power of two
	const int J = tid & (n - 1), K = tid >> lg;
	const int j = n - J, k = n - K;
	double * srow = s + (ssy*j + ssz*k);
	const double * rrow = rhs + (ssy*j + ssz*k);
	const bool mine = tid < nface;
	for (int plane = 0; plane < nplanes; plane++) {
	  const int I = plane - J - K;
	  if (mine && (unsigned) I < (unsigned) n) {
	    const int i = I + 1;
	    double b = 0.;
	    b += 1.*srow[i + 1];
	    b += 1.*srow[i - 1];
	    b += 1.*srow[i + ssy];
	    b += 1.*srow[i - ssy];
	    b += 1.*srow[i + ssz];
	    b += 1.*srow[i - ssz];
	    srow[i] = divide_by_6 (b - rrow[i]);
	  }
	  __syncthreads ();
	}
      }
      else
	for (int plane = 0; plane < nplanes; plane++) {
	  for (int t = tid; t < nface; t += nt) {
	    int J = t % n, K = DIM == 3 ? t / n : 0;
	    int I = plane - J - K;
	    if (I >= 0 && I < n) {
	      int i = I + 1, j = n - J, k = DIM == 3 ? n - K : 0;
	      long c = i + ssy*j + ssz*k;
	      if (DIM == 3 && A.dimension == 3)
		s[c] = relax_value_six (s, c, ssy, ssz, rhs[c]);
	      else
		s[c] = relax_value<DIM, 0> (s, c, ssy, ssz, rhs[c], 0., A.dimension, A.omega, 1., 1.);
	    }
	  }
	  __syncthreads ();
	}
    }
    // natural copy of the own box with the ghost layer of the last BC application
    if (own) {
      const int nall = DIM == 3 ? r*r*r : r*r;
      for (int q = tid; q < nall; q += nt) {
	int i = q % r, j = (q / r) % r, k = DIM == 3 ? q / (r*r) : 0;
	A.dp[l][L.idx (i, j, k)] = s[q];
      }
    }
    __syncthreads ();
  }
}

// the "a barrier wait gave up" word: host memory mapped on the device, next to the words of the
// sweep kernels (relax_skew.hip: h_pinned + 32, one per level; this one after them)
static unsigned * lattice_err_word (gfship_domain * dom)
{
  return (unsigned *) (dom->h_pinned + 32) + GFSHIP_MAXLEVEL + 2;
}

static int lattice_attr (gfship_domain * dom)
{
  if (!dom->lattice_attr_set) {
    GFSHIP_HIP (hipFuncSetAttribute ((const void *) lattice_cycle_kernel<3>,
				     hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
    GFSHIP_HIP (hipFuncSetAttribute ((const void *) lattice_cycle_kernel<2>,
				     hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
    dom->lattice_attr_set = true;
  }
  return GFSHIP_OK;
}

// highest level of the replicated coarse end (lmin .. ltop in LDS per box); -1 when it does not apply
int lattice_cycle_top (gfship_domain * dom, int minlevel, Field * dia)
{
  if (!dom->has_external || dom->no_lattice_cycle || dom->overlap || dom->weighted)
    return -1;
  if (!(dom->comm || dom->gather) || dom->lat_rank < 0 || dom->lat_n < 1 || dom->lat_n > LAT_MAXBOXES)
    return -1;
  if (dom->relax_mode != GFSHIP_RELAX_EXACT || dom->force_hyperplane || dom->no_fused_loop)
    return -1;
  int top = -1;
  size_t bytes = 0, bytes_top = 0;
  for (int l = minlevel; l < dom->depth; l++) {     /* the leaf level is never part of it */
    if (dom->dim == 3 && skew_supported (dom, l)) break;
    if (!dia->zero[l]) break;
    size_t r = dom->lay[l].n + 2;
    bytes += 2*(dom->dim == 3 ? r*r*r : r*r)*sizeof (double);
    if (bytes > 150*1024) break;
    top = l;
    bytes_top = bytes;
  }
  if (top >= 0) {
    /* the workgroups meet at a barrier: all lat_n of them must be resident at once (1024 threads and
       up to 150 KB of LDS each: one per CU).  Checked against the occupancy of the kernel on this
       device; a device that does not deliver (CUs masked or in use) is caught by the bounded wait */
    if (int r = lattice_attr (dom)) { (void) r; return -1; }
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = dom->dim == 3 ?
      hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, lattice_cycle_kernel<3>, 1024, bytes_top) :
      hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, lattice_cycle_kernel<2>, 1024, bytes_top);
    if (e != hipSuccess || hipGetDevice (&dev) != hipSuccess ||
	hipGetDeviceProperties (&prop, dev) != hipSuccess ||
	(long) per_cu*prop.multiProcessorCount < dom->lat_n)
      return -1;
  }
  return top;
}

int lattice_check_error (gfship_domain * dom)
{
  unsigned * w = lattice_err_word (dom);
  if (*w != 0) {
    *w = 0;
    dom->no_lattice_cycle = true;
    set_error ("lattice_cycle_kernel: a barrier wait timed out (workgroups not all resident: CUs held "
	       "by another process or stream?); this solve failed, the domain falls back to one "
	       "exchange per sweep and level");
    return GFSHIP_EHIP;
  }
  return GFSHIP_OK;
}

int launch_lattice_cycle (gfship_domain * dom, unsigned dimension, double omega, int lmin, int ltop,
			  const unsigned * nrelax, Field * dp, Field * ubc, Field * res)
{
  LatticeCycleArgs A;
  const int B = dom->lat_n;
  const Layout & Lt = dom->lay[ltop];
  const size_t xface = dom->dim == 3 ? (size_t) Lt.n*Lt.n : (size_t) Lt.n;
  const size_t need_res = (size_t) B*Lt.total, need_xch = (size_t) 2*B*6*xface;
  if (dom->lat_res_doubles < need_res) {
    if (dom->lat_res) GFSHIP_HIP (hipFree (dom->lat_res));
    dom->lat_res = nullptr; dom->lat_res_doubles = 0;
    GFSHIP_HIP (hipMalloc ((void **) &dom->lat_res, need_res*sizeof (double)));
    dom->lat_res_doubles = need_res;
  }
  if (dom->lat_xch_doubles < need_xch) {
    if (dom->lat_xch) GFSHIP_HIP (hipFree (dom->lat_xch));
    dom->lat_xch = nullptr; dom->lat_xch_doubles = 0;
    GFSHIP_HIP (hipMalloc ((void **) &dom->lat_xch, need_xch*sizeof (double)));
    dom->lat_xch_doubles = need_xch;
  }
  if (!dom->lat_bar)
    GFSHIP_HIP (hipMalloc ((void **) &dom->lat_bar, 64));
  // MPI_Allgather of the residual of the level ltop (whole arrays: contiguous, no pack kernel)
  int r = call_gather (dom, res->lev[ltop], dom->lat_res, Lt.total);
  if (r) return r;
  GFSHIP_HIP (hipMemsetAsync (dom->lat_bar, 0, 64, dom->stream));
  size_t bytes = 0;
  for (int l = 0; l <= GFSHIP_MAXLEVEL; l++) {
    A.lay[l] = dom->lay[l <= dom->depth ? l : dom->depth];
    A.nrelax[l] = 0; A.dp[l] = nullptr; A.res[l] = nullptr;
  }
  for (int l = lmin; l <= ltop; l++) {
    size_t rr = dom->lay[l].n + 2;
    bytes += 2*(dom->dim == 3 ? rr*rr*rr : rr*rr)*sizeof (double);
    A.nrelax[l] = nrelax[l];
    A.dp[l] = dp->lev[l];
    A.res[l] = res->lev[l];
    dp->zero[l] = false;
    res->zero[l] = false;
  }
  A.gres = dom->lat_res;
  A.xch = (unsigned long long *) dom->lat_xch;
  A.bar = dom->lat_bar;
  A.lmin = lmin; A.ltop = ltop;
  for (int d = 0; d < 6; d++) {
    A.bc.side[d] = dom->side[d];
    A.bc.type[d] = ubc->bc[d];
    A.bc.val[d] = nullptr;
  }
  A.bc.component = ubc->component;
  A.bc.homogeneous = 1;
  A.dimension = dimension;
  A.omega = omega;
  A.rank = dom->lat_rank; A.nboxes = B;
  for (int c = 0; c < 3; c++) A.b[c] = dom->lat_b[c];
  A.err = lattice_err_word (dom);
  if ((r = lattice_attr (dom))) return r;
  if (dom->dim == 3)
    hipLaunchKernelGGL (lattice_cycle_kernel<3>, dim3 (B), dim3 (1024), bytes, dom->stream, A);
  else
    hipLaunchKernelGGL (lattice_cycle_kernel<2>, dim3 (B), dim3 (1024), bytes, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  dom->n_lattice_cycles++;
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Red-black Gauss-Seidel of the same operator (opt-in, not a reference algorithm): two colour
// passes, each cell updated from the current values of its six neighbours.
// ---------------------------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(256)
relax_redblack_kernel (Layout L, int colour, unsigned dimension, double omega,
		       double * __restrict__ u, const double * __restrict__ rhs,
		       const double * __restrict__ dia)
{
  int n = L.n;
  int half = (n + 1)/2;
  int ih = blockIdx.x*blockDim.x + threadIdx.x;
  int j = blockIdx.y + 1;
  int k = DIM == 3 ? blockIdx.z + 1 : 0;
  if (ih >= half) return;
  int i = 2*ih + 1 + ((j + k + colour) & 1);
  if (i > n) return;
  long c = L.idx (i, j, k);
  u[c] = relax_value<DIM, 0> (u, c, L.sy, L.sz, rhs[c], dia[c], dimension, omega, 1., 1.);
}

int launch_relax_redblack (gfship_domain * dom, unsigned dimension, int level, double omega,
			   double * u, const double * rhs, const double * dia)
{
  const Layout & L = dom->lay[level];
  int half = (L.n + 1)/2;
  int block = half >= 128 ? 128 : 64;
  dim3 grid ((half + block - 1)/block, L.n, dom->dim == 3 ? L.n : 1);
  for (int colour = 0; colour < 2; colour++) {
    if (dom->dim == 3)
      hipLaunchKernelGGL (relax_redblack_kernel<3>, grid, dim3 (block), 0, dom->stream,
			  L, colour, dimension, omega, u, rhs, dia);
    else
      hipLaunchKernelGGL (relax_redblack_kernel<2>, grid, dim3 (block), 0, dom->stream,
			  L, colour, dimension, omega, u, rhs, dia);
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Cell-parallel kernels: one thread per interior cell, x fastest.
// ---------------------------------------------------------------------------------------------
#define CELL_LOOP_PROLOGUE(L)						\
  int i = blockIdx.x*blockDim.x + threadIdx.x + 1;			\
  int j = blockIdx.y + 1;						\
  int k = (L).dim == 3 ? blockIdx.z + 1 : 0;				\
  if (i > (L).n) return;						\
  long c = (L).idx (i, j, k)

static inline void cell_grid (const Layout & L, dim3 * grid, dim3 * block)
{
  int b = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
  *block = dim3 (b);
  *grid = dim3 ((L.n + b - 1)/b, L.n, L.dim == 3 ? L.n : 1);
}

// K2: residual_set / residual_set2D, src/poisson.c:634-678
template <int DIM>
__global__ void __launch_bounds__(256)
residual_kernel (Layout L, const double * __restrict__ u, const double * __restrict__ rhs,
		 const double * __restrict__ dia, double * __restrict__ res)
{
  CELL_LOOP_PROLOGUE (L);
  double a = dia[c], b = 0.;
  a += 1.; b += 1.*u[c + 1];
  a += 1.; b += 1.*u[c - 1];
  a += 1.; b += 1.*u[c + L.sy];
  a += 1.; b += 1.*u[c - L.sy];
  if (DIM == 3) {
    a += 1.; b += 1.*u[c + L.sz];
    a += 1.; b += 1.*u[c - L.sz];
  }
  res[c] = rhs[c] - (b - u[c]*a);
}

template <bool NORM>
__global__ void residual_norm2_kernel (Layout L, const double * __restrict__ u, const double * __restrict__ rhs,
				       const double * __restrict__ dia, double * __restrict__ res, double inv,
				       double weight, double * __restrict__ partial);

int launch_residual (gfship_domain * dom, int level, const double * u, const double * rhs,
		     const double * dia, double * res)
{
  const Layout & L = dom->lay[level];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  static const bool pairs = getenv ("GFSHIP_RN_SCALAR") == nullptr;
  if (dom->dim == 3 && L.n >= 64 && pairs) {
    /* two cells per thread, 16-byte accesses (residual_norm2_kernel without the norm) */
    const long nitems = (long) L.n*L.n*(L.n/2);
    long nb = (nitems + 255)/256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL (residual_norm2_kernel<false>, dim3 ((int) nb), dim3 (256), 0, dom->stream, L, u, rhs,
			dia, res, 1., 1., nullptr);
    GFSHIP_HIP (hipGetLastError ());
    return GFSHIP_OK;
  }
  if (dom->dim == 3)
    hipLaunchKernelGGL (residual_kernel<3>, grid, block, 0, dom->stream, L, u, rhs, dia, res);
  else
    hipLaunchKernelGGL (residual_kernel<2>, grid, block, 0, dom->stream, L, u, rhs, dia, res);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// K2 with the face weights of the cells (residual_set, src/poisson.c:634-655)
template <int DIM>
__global__ void __launch_bounds__(256)
residual_weighted_kernel (Layout L, const double * __restrict__ u, const double * __restrict__ rhs,
			  const double * __restrict__ dia, double * __restrict__ res, W6 wf)
{
  CELL_LOOP_PROLOGUE (L);
  double a = dia[c], b = 0.;
  { const double g = wf.p[0][c]; a += g; b += g*u[c + 1]; }
  { const double g = wf.p[1][c]; a += g; b += g*u[c - 1]; }
  { const double g = wf.p[2][c]; a += g; b += g*u[c + L.sy]; }
  { const double g = wf.p[3][c]; a += g; b += g*u[c - L.sy]; }
  if (DIM == 3) {
    { const double g = wf.p[4][c]; a += g; b += g*u[c + L.sz]; }
    { const double g = wf.p[5][c]; a += g; b += g*u[c - L.sz]; }
  }
  res[c] = rhs[c] - (b - u[c]*a);
}

RelaxOp weighted_op (gfship_domain * dom, int level)
{
  RelaxOp op;
  op.kind = 2;
  for (int d = 0; d < 2*dom->dim; d++)
    op.wf[d] = dom->fields[dom->wf[d]].lev[level];
  return op;
}

int launch_residual_weighted (gfship_domain * dom, int level, const double * u, const double * rhs,
			      const double * dia, double * res)
{
  const Layout & L = dom->lay[level];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  RelaxOp op = weighted_op (dom, level);
  W6 wf;
  for (int d = 0; d < 6; d++) wf.p[d] = op.wf[d];
  if (dom->dim == 3)
    hipLaunchKernelGGL (residual_weighted_kernel<3>, grid, block, 0, dom->stream, L, u, rhs, dia, res, wf);
  else
    hipLaunchKernelGGL (residual_weighted_kernel<2>, grid, block, 0, dom->stream, L, u, rhs, dia, res, wf);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// K8: gfs_poisson_coefficients (src/poisson.c:856-901) with a GfsFunction alpha given at the faces.
//   leaves (reset_coeff :756-767 + poisson_coeff :769-797): every face adds v = lambda2*alpha*
//     face_fraction/metric = 1.*alpha*1./1. once to f[d].v of the cells on both sides, i.e.
//     f[d].v = 0. + v; alpha[c] holds the face values in the layout of a level array (entry of a
//     cell = its + face along c, the entry in front of the first cell = its - face);
//   non-leaf cells, finest first (face_coeff_from_below :826-853): the mean over the children on
//     side d in child-id order (ftt_cell_children_direction, src/ftt.h:322-341), then all six
//     zeroed if exactly one direction has a non-zero weight towards a cell of the box.
struct A3 { const double * p[3]; };
struct M6 { double * p[6]; };

template <int DIM>
__global__ void __launch_bounds__(256)
weights_leaf_kernel (Layout L, A3 alpha, M6 w)
{
  CELL_LOOP_PROLOGUE (L);
  const long off[3] = { 1, L.sy, L.sz };
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    double v = 1.*alpha.p[cc][c]*1./1.;
    double f = 0.;
    f += v;
    w.p[2*cc][c] = f;
    v = 1.*alpha.p[cc][c - off[cc]]*1./1.;
    f = 0.;
    f += v;
    w.p[2*cc + 1][c] = f;
  }
}

template <int DIM>
__global__ void __launch_bounds__(256)
weights_coarse_kernel (Layout Lc, Layout Lf, M6 wc, W6 wfine)
{
  CELL_LOOP_PROLOGUE (Lc);
  constexpr int ND = DIM == 3 ? 4 : 2;
  const int cdir3[6][4] = {{1,3,5,7},{0,2,4,6},{0,1,4,5},{2,3,6,7},{0,1,2,3},{4,5,6,7}};
  const int cdir2[4][2] = {{1,3},{0,2},{0,1},{2,3}};
  const int n = Lc.n;
  const int ijk[3] = { i, j, k };
  double s6[6];
  unsigned neighbors = 0;
#pragma unroll
  for (int d = 0; d < 2*DIM; d++) {
    double s = 0.;
#pragma unroll
    for (int m = 0; m < ND; m++) {
      const int id = DIM == 3 ? cdir3[d][m] : cdir2[d][m];
      const int ci = 2*i - 1 + (id & 1);
      const int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
      const int ck = DIM == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
      s += wfine.p[d][Lf.idx (ci, cj, ck)];
    }
    s /= ND;
    s6[d] = s;
    const bool at_side = (d & 1) ? ijk[d/2] == 1 : ijk[d/2] == n;
    if (s != 0. && !at_side)
      neighbors++;
  }
#pragma unroll
  for (int d = 0; d < 2*DIM; d++)
    wc.p[d][c] = neighbors == 1 ? 0. : s6[d];
}

int launch_poisson_weights (gfship_domain * dom, double * const alpha[3])
{
  const int Ld = dom->depth;
  {
    const Layout & L = dom->lay[Ld];
    dim3 grid, block;
    cell_grid (L, &grid, &block);
    A3 a;
    M6 w;
    for (int c = 0; c < 3; c++) a.p[c] = c < dom->dim ? alpha[c] : nullptr;
    for (int d = 0; d < 6; d++) w.p[d] = d < 2*dom->dim ? dom->fields[dom->wf[d]].lev[Ld] : nullptr;
    if (dom->dim == 3) hipLaunchKernelGGL (weights_leaf_kernel<3>, grid, block, 0, dom->stream, L, a, w);
    else               hipLaunchKernelGGL (weights_leaf_kernel<2>, grid, block, 0, dom->stream, L, a, w);
  }
  for (int l = Ld - 1; l >= 0; l--) {
    const Layout & Lc = dom->lay[l], & Lf = dom->lay[l + 1];
    dim3 grid, block;
    cell_grid (Lc, &grid, &block);
    M6 wc;
    W6 wfine;
    for (int d = 0; d < 6; d++) {
      wc.p[d] = d < 2*dom->dim ? dom->fields[dom->wf[d]].lev[l] : nullptr;
      wfine.p[d] = d < 2*dom->dim ? dom->fields[dom->wf[d]].lev[l + 1] : nullptr;
    }
    if (dom->dim == 3) hipLaunchKernelGGL (weights_coarse_kernel<3>, grid, block, 0, dom->stream, Lc, Lf, wc, wfine);
    else               hipLaunchKernelGGL (weights_coarse_kernel<2>, grid, block, 0, dom->stream, Lc, Lf, wc, wfine);
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// K3: get_from_below_3D / _2D, src/poisson.c:1044-1068: sum of the children in child-id order
// (child id bit0 -> +x, bit1 -> -y, bit2 -> -z), halved when dimension == 3.
template <int DIM>
__global__ void __launch_bounds__(256)
restrict_kernel (Layout Lc, Layout Lf, unsigned dimension, double * __restrict__ vc,
		 const double * __restrict__ vf)
{
  CELL_LOOP_PROLOGUE (Lc);
  double val = 0.;
#pragma unroll
  for (int id = 0; id < (1 << DIM); id++) {
    int ci = 2*i - 1 + (id & 1);
    int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
    int ck = DIM == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
    val += vf[Lf.idx (ci, cj, ck)];
  }
  vc[c] = dimension == 2 ? val : val/2.;
}

int launch_restrict (gfship_domain * dom, unsigned dimension, int level_coarse, double * v_coarse,
		     const double * v_fine)
{
  const Layout & Lc = dom->lay[level_coarse], & Lf = dom->lay[level_coarse + 1];
  dim3 grid, block;
  cell_grid (Lc, &grid, &block);
  if (dom->dim == 3)
    hipLaunchKernelGGL (restrict_kernel<3>, grid, block, 0, dom->stream, Lc, Lf, dimension,
			v_coarse, v_fine);
  else
    hipLaunchKernelGGL (restrict_kernel<2>, grid, block, 0, dom->stream, Lc, Lf, dimension,
			v_coarse, v_fine);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// K4: get_from_above, src/poisson.c:1005-1042, one thread per child (fine cell).
// gfs_face_gradient at max_level = parent level (src/fluid.c:801-805): g.a = 1., g.b = nb.
template <int DIM>
__global__ void __launch_bounds__(256)
prolongate_kernel (Layout Lc, Layout Lf, const double * __restrict__ vc,
		   double * __restrict__ vf)
{
  CELL_LOOP_PROLOGUE (Lf);
  int pi = (i + 1)/2, pj = (j + 1)/2, pk = DIM == 3 ? (k + 1)/2 : 0;
  long p = Lc.idx (pi, pj, pk);
  double pv = vc[p];
  double h[3];
  const long off[3] = { 1, Lc.sy, Lc.sz };
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    double g1 = vc[p + off[cc]] - 1.*pv;
    double g2 = vc[p - off[cc]] - 1.*pv;
    h[cc] = (g1 - g2)/2.;
  }
  // relative position of the child, ftt_cell_relative_pos: coords[n]/4. (src/ftt.c:327-340)
  double rel[3] = { ((i & 1) ? -1. : 1.)/4., ((j & 1) ? -1. : 1.)/4., ((k & 1) ? -1. : 1.)/4. };
  double val = pv;
#pragma unroll
  for (int cc = 0; cc < DIM; cc++)
    val += rel[cc]*h[cc];
  vf[c] = val;
}

int launch_prolongate (gfship_domain * dom, int level_coarse, const double * v_coarse,
		       double * v_fine)
{
  const Layout & Lc = dom->lay[level_coarse], & Lf = dom->lay[level_coarse + 1];
  dim3 grid, block;
  cell_grid (Lf, &grid, &block);
  if (dom->dim == 3)
    hipLaunchKernelGGL (prolongate_kernel<3>, grid, block, 0, dom->stream, Lc, Lf, v_coarse, v_fine);
  else
    hipLaunchKernelGGL (prolongate_kernel<2>, grid, block, 0, dom->stream, Lc, Lf, v_coarse, v_fine);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// K5: correct, src/poisson.c:998-1003
__global__ void __launch_bounds__(256)
correct_kernel (Layout L, double * __restrict__ u, const double * __restrict__ dp)
{
  CELL_LOOP_PROLOGUE (L);
  u[c] += dp[c];
}

int launch_correct (gfship_domain * dom, int level, double * u, const double * dp)
{
  const Layout & L = dom->lay[level];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  hipLaunchKernelGGL (correct_kernel, grid, block, 0, dom->stream, L, u, dp);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// K6: gfs_cell_reset (src/fluid.c:1995) and general fill, ghosts included
__global__ void fill_kernel (double * __restrict__ a, size_t total, double value)
{
  size_t q = (size_t) blockIdx.x*blockDim.x + threadIdx.x;
  size_t stride = (size_t) gridDim.x*blockDim.x;
  for (; q < total; q += stride)
    a[q] = value;
}

int launch_fill (gfship_domain * dom, int level, double * a, double value)
{
  const Layout & L = dom->lay[level];
  int block = 256;
  size_t g = (L.total + block - 1)/block;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL (fill_kernel, dim3 ((unsigned) g), dim3 (block), 0, dom->stream, a, L.total, value);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// K7: norms.  gfs_norm_add (src/fluid.c:2139-2154) per cell on val = a[c]/scale with weight w;
// the sums are tree-reduced (deterministic, but not the reference's sequential order: they
// agree to rounding); infty = max |val| is exact.  out = { sum w*val, sum w*|val|,
// sum w*|val|*|val|, max |val|, sum a[c] }.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum (double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_down (v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max (double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v = fmax (v, __shfl_down (v, o, 64));
  return v;
}

// rows of cells per workgroup (grid-stride), lanes along x: no per-cell index arithmetic.  EXACT_INV:
// scale is a power of two (h*h), the division raw/scale is the exact multiplication by 1/scale.
template <bool EXACT_INV>
__global__ void __launch_bounds__(256)
norm_partial_kernel (Layout L, const double * __restrict__ a, double scale, double weight,
		     double * __restrict__ partial)
{
  const int n = L.n;
  const long nrows = L.dim == 3 ? (long) n*n : n;
  const double inv = 1./scale;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0., s4 = 0.;
  for (long r = blockIdx.x; r < nrows; r += gridDim.x) {
    const int j = (int) (r % n) + 1, k = L.dim == 3 ? (int) (r / n) + 1 : 0;
    const double * row = a + L.idx (1, j, k);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      double raw = row[i];
      double val = EXACT_INV ? raw*inv : raw/scale;
      s0 += weight*val;
      val = fabs (val);
      s3 = fmax (s3, val);
      s1 += weight*val;
      s2 += weight*val*val;
      s4 += raw;
    }
  }
  __shared__ double sh[5][4];
  s0 = wave_sum (s0); s1 = wave_sum (s1); s2 = wave_sum (s2); s3 = wave_max (s3); s4 = wave_sum (s4);
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = s3; sh[4][w] = s4; }
  __syncthreads ();
  if (threadIdx.x == 0) {
    int nw = blockDim.x >> 6;
    double r0 = 0., r1 = 0., r2 = 0., r3 = 0., r4 = 0.;
    for (int q = 0; q < nw; q++) {
      r0 += sh[0][q]; r1 += sh[1][q]; r2 += sh[2][q]; r3 = fmax (r3, sh[3][q]); r4 += sh[4][q];
    }
    double * p = partial + 5*(size_t) blockIdx.x;
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3; p[4] = r4;
  }
}

__global__ void __launch_bounds__(256)
norm_final_kernel (const double * __restrict__ partial, int nblocks, double * __restrict__ out)
{
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0., s4 = 0.;
  for (int q = threadIdx.x; q < nblocks; q += blockDim.x) {
    const double * p = partial + 5*(size_t) q;
    s0 += p[0]; s1 += p[1]; s2 += p[2]; s3 = fmax (s3, p[3]); s4 += p[4];
  }
  __shared__ double sh[5][4];
  s0 = wave_sum (s0); s1 = wave_sum (s1); s2 = wave_sum (s2); s3 = wave_max (s3); s4 = wave_sum (s4);
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = s3; sh[4][w] = s4; }
  __syncthreads ();
  if (threadIdx.x == 0) {
    double r0 = 0., r1 = 0., r2 = 0., r3 = 0., r4 = 0.;
    for (int q = 0; q < 4; q++) {
      r0 += sh[0][q]; r1 += sh[1][q]; r2 += sh[2][q]; r3 = fmax (r3, sh[3][q]); r4 += sh[4][q];
    }
    out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; out[4] = r4;
  }
}

// 1/scale is exact when scale is a power of two (the h*h of the residual norms, 1. elsewhere)
static bool power_of_two (double x)
{
  int e;
  return x > 0. && frexp (x, &e) == 0.5;
}
#define NORM_PARTIAL_LAUNCH() do {					\
    if (power_of_two (scale))						\
      hipLaunchKernelGGL (norm_partial_kernel<true>, dim3 (nblocks), dim3 (block), 0, dom->stream, \
			  L, a, scale, weight, partial);		\
    else								\
      hipLaunchKernelGGL (norm_partial_kernel<false>, dim3 (nblocks), dim3 (block), 0, dom->stream, \
			  L, a, scale, weight, partial);		\
  } while (0)

// the same without waiting for the result: it lands in dom->h_pinned[8 .. 12] once the stream has
// passed this point (the caller synchronises later anyway)
int launch_norm_async (gfship_domain * dom, int level, const double * a, double scale, double weight)
{
  const Layout & L = dom->lay[level];
  long nrows = L.dim == 3 ? (long) L.n*L.n : L.n;
  int block = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
  int nblocks = (int) (nrows > 1024 ? 1024 : nrows);
  double * partial = dom->d_scratch;
  double * result = dom->h_pinned + 8;     /* host memory mapped on the device */
  NORM_PARTIAL_LAUNCH ();
  hipLaunchKernelGGL (norm_final_kernel, dim3 (1), dim3 (256), 0, dom->stream,
		      partial, nblocks, result);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int launch_norm (gfship_domain * dom, int level, const double * a, double scale, double weight,
		 double out[5])
{
  const Layout & L = dom->lay[level];
  long nrows = L.dim == 3 ? (long) L.n*L.n : L.n;
  int block = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
  int nblocks = (int) (nrows > 1024 ? 1024 : nrows);
  double * partial = dom->d_scratch;     // 5*1024 doubles
  double * result = dom->h_pinned;       // host memory mapped on the device: no copy kernel
  NORM_PARTIAL_LAUNCH ();
  hipLaunchKernelGGL (norm_final_kernel, dim3 (1), dim3 (256), 0, dom->stream,
		      partial, nblocks, result);
  GFSHIP_HIP (hipGetLastError ());
  GFSHIP_HIP (stream_wait_spin (dom->stream));
  memcpy (out, dom->h_pinned, 5*sizeof (double));
  return GFSHIP_OK;
}

// K2 + K7 in one pass: the residual of the leaves and the partial sums of its norm
// (gfs_residual src/poisson.c:721-747 followed by gfs_domain_norm_residual src/domain.c:2264-2288:
// the solve loop always asks for both).  Rows of cells per workgroup as in norm_partial_kernel, the
// residual arithmetic of residual_kernel; the residual is stored and accumulated from the register.
// scale = h*h is a power of two: the division by it is the exact multiplication by 1/scale.
template <int DIM>
__global__ void __launch_bounds__(256)
residual_norm_kernel (Layout L, const double * __restrict__ u, const double * __restrict__ rhs,
		      const double * __restrict__ dia, double * __restrict__ res, double inv,
		      double weight, double * __restrict__ partial)
{
  const int n = L.n;
  const long nrows = DIM == 3 ? (long) n*n : n;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0., s4 = 0.;
  for (long r = blockIdx.x; r < nrows; r += gridDim.x) {
    const int j = (int) (r % n) + 1, k = DIM == 3 ? (int) (r / n) + 1 : 0;
    const long row = L.idx (1, j, k);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const long c = row + i;
      double a = dia ? dia[c] : 0., b = 0.;     /* dia == nullptr: the level holds zeros (not read) */
      a += 1.; b += 1.*u[c + 1];
      a += 1.; b += 1.*u[c - 1];
      a += 1.; b += 1.*u[c + L.sy];
      a += 1.; b += 1.*u[c - L.sy];
      if (DIM == 3) {
	a += 1.; b += 1.*u[c + L.sz];
	a += 1.; b += 1.*u[c - L.sz];
      }
      const double raw = rhs[c] - (b - u[c]*a);
      res[c] = raw;
      double val = raw*inv;
      s0 += weight*val;
      val = fabs (val);
      s3 = fmax (s3, val);
      s1 += weight*val;
      s2 += weight*val*val;
      s4 += raw;
    }
  }
  __shared__ double sh[5][4];
  s0 = wave_sum (s0); s1 = wave_sum (s1); s2 = wave_sum (s2); s3 = wave_max (s3); s4 = wave_sum (s4);
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = s3; sh[4][w] = s4; }
  __syncthreads ();
  if (threadIdx.x == 0) {
    int nw = blockDim.x >> 6;
    double r0 = 0., r1 = 0., r2 = 0., r3 = 0., r4 = 0.;
    for (int q = 0; q < nw; q++) {
      r0 += sh[0][q]; r1 += sh[1][q]; r2 += sh[2][q]; r3 = fmax (r3, sh[3][q]); r4 += sh[4][q];
    }
    double * p = partial + 5*(size_t) blockIdx.x;
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3; p[4] = r4;
  }
}

// The same for 3-D levels with n >= 4, two cells of a row per thread: the rows of a level start 16-byte
// aligned (Layout), so the centre, the four neighbour rows, the right-hand side and the result move
// as 16-byte accesses and only the two cells beside the pair as 8-byte ones -- 8 loads and 1 store per
// two cells instead of 16 and 2 (the 8-byte version moved 3.1 TB/s of its 24 B per cell at 256^3: bound
// by the number of requests, not by bytes).  Same arithmetic per cell; the sums are accumulated in
// another order (they are tree-reduced anyway: 1e-12), the maximum is exact.
template <bool NORM>
__global__ void __launch_bounds__(256)
residual_norm2_kernel (Layout L, const double * __restrict__ u, const double * __restrict__ rhs,
		       const double * __restrict__ dia, double * __restrict__ res, double inv,
		       double weight, double * __restrict__ partial)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int n = L.n, half = n >> 1;
  const long nitems = (long) n*n*half;
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0., s4 = 0.;
  for (long it = (long) blockIdx.x*blockDim.x + threadIdx.x; it < nitems; it += (long) gridDim.x*blockDim.x) {
    const int pr = (int) (it % half);
    const long r = it / half;
    const int j = (int) (r % n) + 1, k = (int) (r / n) + 1;
    const long c = L.idx (1 + 2*pr, j, k);           /* even index: 16-byte aligned */
    const d2 uc = *(const d2 *) (u + c);
    const double ul = u[c - 1], ur = u[c + 2];
    const d2 ut = *(const d2 *) (u + c + L.sy), ub = *(const d2 *) (u + c - L.sy);
    const d2 uf = *(const d2 *) (u + c + L.sz), uk = *(const d2 *) (u + c - L.sz);
    const d2 rh = *(const d2 *) (rhs + c);
    d2 di = { 0., 0. };
    if (dia) di = *(const d2 *) (dia + c);
    d2 out;
    {
      double a = di.x, b = 0.;
      a += 1.; b += 1.*uc.y;
      a += 1.; b += 1.*ul;
      a += 1.; b += 1.*ut.x;
      a += 1.; b += 1.*ub.x;
      a += 1.; b += 1.*uf.x;
      a += 1.; b += 1.*uk.x;
      out.x = rh.x - (b - uc.x*a);
    }
    {
      double a = di.y, b = 0.;
      a += 1.; b += 1.*ur;
      a += 1.; b += 1.*uc.x;
      a += 1.; b += 1.*ut.y;
      a += 1.; b += 1.*ub.y;
      a += 1.; b += 1.*uf.y;
      a += 1.; b += 1.*uk.y;
      out.y = rh.y - (b - uc.y*a);
    }
    *(d2 *) (res + c) = out;
    if (NORM)
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const double raw = e ? out.y : out.x;
      double val = raw*inv;
      s0 += weight*val;
      val = fabs (val);
      s3 = fmax (s3, val);
      s1 += weight*val;
      s2 += weight*val*val;
      s4 += raw;
    }
  }
  if (!NORM) return;
  __shared__ double sh[5][4];
  s0 = wave_sum (s0); s1 = wave_sum (s1); s2 = wave_sum (s2); s3 = wave_max (s3); s4 = wave_sum (s4);
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = s3; sh[4][w] = s4; }
  __syncthreads ();
  if (threadIdx.x == 0) {
    int nw = blockDim.x >> 6;
    double r0 = 0., r1 = 0., r2 = 0., r3 = 0., r4 = 0.;
    for (int q = 0; q < nw; q++) {
      r0 += sh[0][q]; r1 += sh[1][q]; r2 += sh[2][q]; r3 = fmax (r3, sh[3][q]); r4 += sh[4][q];
    }
    double * p = partial + 5*(size_t) blockIdx.x;
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3; p[4] = r4;
  }
}

// result_slot: 0 = dom->h_pinned (waited for, copied to out), 8 = dom->h_pinned + 8 (not waited for:
// the caller synchronises later, launch_norm_async's convention)
int launch_residual_norm (gfship_domain * dom, int level, const double * u, const double * rhs,
			  const double * dia, double * res, double scale, double weight,
			  double * out, bool dia_zero)
{
  const Layout & L = dom->lay[level];
  if (!power_of_two (scale)) {
    int r = launch_residual (dom, level, u, rhs, dia, res);
    if (r) return r;
    return out ? launch_norm (dom, level, res, scale, weight, out) :
      launch_norm_async (dom, level, res, scale, weight);
  }
  long nrows = L.dim == 3 ? (long) L.n*L.n : L.n;
  int block = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
  /* d_scratch holds 5*8192 partials; GFSHIP_RN_BLOCKS: tuning knob (4096 = 16 workgroups per CU:
     the 7-point reads of a row keep more rows in flight than the norm alone needs) */
  static int rn_blocks = 0;
  if (!rn_blocks) {
    const char * e = getenv ("GFSHIP_RN_BLOCKS");
    rn_blocks = e ? atoi (e) : 4096;
    if (rn_blocks < 64 || rn_blocks > 8192) rn_blocks = 4096;
  }
  int nblocks = (int) (nrows > rn_blocks ? rn_blocks : nrows);
  double * partial = dom->d_scratch;
  double * result = out ? dom->h_pinned : dom->h_pinned + 8;
  static const bool rn_pairs = getenv ("GFSHIP_RN_SCALAR") == nullptr;
  if (dom->dim == 3 && L.n >= 64 && rn_pairs) {
    const long nitems = (long) L.n*L.n*(L.n/2);
    long nb = (nitems + 255)/256;
    nblocks = (int) (nb > rn_blocks ? rn_blocks : nb);
    hipLaunchKernelGGL (residual_norm2_kernel<true>, dim3 (nblocks), dim3 (256), 0, dom->stream, L, u,
			rhs, dia_zero ? nullptr : dia, res, 1./scale, weight, partial);
  }
  else if (dom->dim == 3)
    hipLaunchKernelGGL (residual_norm_kernel<3>, dim3 (nblocks), dim3 (block), 0, dom->stream, L, u,
			rhs, dia_zero ? nullptr : dia, res, 1./scale, weight, partial);
  else
    hipLaunchKernelGGL (residual_norm_kernel<2>, dim3 (nblocks), dim3 (block), 0, dom->stream, L, u,
			rhs, dia_zero ? nullptr : dia, res, 1./scale, weight, partial);
  hipLaunchKernelGGL (norm_final_kernel, dim3 (1), dim3 (256), 0, dom->stream,
		      partial, nblocks, result);
  GFSHIP_HIP (hipGetLastError ());
  if (out) {
    GFSHIP_HIP (stream_wait_spin (dom->stream));
    memcpy (out, dom->h_pinned, 5*sizeof (double));
  }
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// The cell order of gfs_traverse_and_homogeneous_bc in a parallel run with the domain parameter
// `overlap = 1', the reference's default (src/domain.c:682,1093-1125): the cells along the
// GfsBoundaryMpi sides first (update_mpi_boundaries :1024-1049: sides d = 0..5 in turn,
// ftt_cell_traverse_boundary in traversal order, a cell only once -- GFS_FLAG_USED), so that their
// new values can travel while the bulk is swept, then the remaining cells in traversal order
// (update_other_cell :1016-1022).
//
// On the device the sweep runs as the levels of the dependency graph of that sequence: cell c
// must follow those of its six neighbours that come before it in the sequence and precede the
// others; every topological order of that relation gives the same bits.  step (c) = 1 + max over
// the earlier neighbours, computed once per level on the host by walking the sequence; the cells
// sorted by step are kept on the device and one launch relaxes the cells of one step.  The steps
// up to `shell_steps' hold every cell along an MPI side (and nothing that depends on the bulk):
// after them the halo layers are final and the exchange can start beside the bulk.
// ---------------------------------------------------------------------------------------------
struct MpiOrderPlan {
  int * cells = nullptr;             // device: linear indices into the level array, sorted by step
  std::vector<int> first;            // first[s] .. first[s+1]: the cells of step s
  int shell_steps = 0;               // steps after which every cell along an MPI side is done
};

template <int DIM, int OP>
__global__ void __launch_bounds__(256)
relax_listed_kernel (const int * __restrict__ cells, int count, long sy, long sz,
		     unsigned dimension, double omega, double w, double h2,
		     double * __restrict__ u, const double * __restrict__ rhs,
		     const double * __restrict__ dia)
{
  const int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= count) return;
  const long c = cells[q];
  u[c] = relax_value<DIM, OP> (u, c, sy, sz, rhs[c], dia[c], dimension, omega, w, h2);
}

// pre-order position of cell (I, J, K) of a level (oriented coordinates: I = i - 1, J = n - j,
// K = n - k): child n of every ancestor = bit 0 from I, bit 1 from J, bit 2 from K
static inline unsigned long long morton_key (int dim, int level, int I, int J, int K)
{
  unsigned long long m = 0;
  for (int b = level - 1; b >= 0; b--) {
    unsigned n = ((I >> b) & 1) | (((J >> b) & 1) << 1) | (dim == 3 ? ((K >> b) & 1) << 2 : 0);
    m = (m << (dim == 3 ? 3 : 2)) | n;
  }
  return m;
}

void mpi_order_free (gfship_domain * dom)
{
  for (int l = 0; l <= GFSHIP_MAXLEVEL; l++)
    if (dom->mpi_plan[l]) {
      MpiOrderPlan * P = (MpiOrderPlan *) dom->mpi_plan[l];
      if (P->cells) (void) hipFree (P->cells);
      delete P;
      dom->mpi_plan[l] = nullptr;
    }
}

static int mpi_order_plan (gfship_domain * dom, int level, MpiOrderPlan ** out)
{
  if (dom->mpi_plan[level]) { *out = (MpiOrderPlan *) dom->mpi_plan[level]; return GFSHIP_OK; }
  const Layout & L = dom->lay[level];
  const int n = L.n, dim = dom->dim;
  const size_t ncell = (size_t) ncells (L);
  // the sequence: cells along the MPI sides, side by side, each in traversal order; then the rest
  std::vector<int> seq;               // linear indices in sequence order
  seq.reserve (ncell);
  std::vector<int> step (L.total, 0); // 0 = not visited yet (ghost cells stay 0)
  std::vector<char> used (L.total, 0);
  size_t nshell = 0;
  for (int s = 0; s < 2*dim; s++) {
    if (dom->side[s] != GFSHIP_SIDE_EXTERNAL) continue;
    // the face: axis c fixed at its first / last cell; sorted by pre-order position
    const int c = s/2, fixed = (s & 1) ? 1 : n;
    std::vector<std::pair<unsigned long long, int>> face;
    const int nb = dim == 3 ? n*n : n;
    face.reserve (nb);
    for (int t = 0; t < nb; t++) {
      int ijk[3] = { 1, 1, dim == 3 ? 1 : 0 };
      int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
      ijk[c] = fixed;
      ijk[ta] = t % n + 1;
      if (dim == 3) ijk[tb] = t / n + 1;
      const long idx = L.idx (ijk[0], ijk[1], ijk[2]);
      if (used[idx]) continue;
      face.emplace_back (morton_key (dim, level, ijk[0] - 1, n - ijk[1], dim == 3 ? n - ijk[2] : 0),
			 (int) idx);
    }
    std::sort (face.begin (), face.end ());
    for (auto & f : face) { used[f.second] = 1; seq.push_back (f.second); }
  }
  nshell = seq.size ();
  {
    // the remaining cells in traversal order: walk the pre-order positions
    const int bits = dim == 3 ? 3 : 2;
    for (unsigned long long m = 0; m < (unsigned long long) ncell; m++) {
      int I = 0, J = 0, K = 0;
      for (int b = 0; b < level; b++) {
	const unsigned nn = (unsigned) ((m >> (bits*b)) & ((1u << bits) - 1));
	I |= (nn & 1) << b; J |= ((nn >> 1) & 1) << b; K |= ((nn >> 2) & 1) << b;
      }
      const long idx = L.idx (I + 1, n - J, dim == 3 ? n - K : 0);
      if (!used[idx]) seq.push_back ((int) idx);
    }
  }
  GFSHIP_CHECK (seq.size () == ncell, GFSHIP_EHIP, "internal error: sweep sequence of %zu cells for %zu",
		seq.size (), ncell);
  // step of every cell: after its earlier neighbours
  const long off[6] = { 1, -1, L.sy, - L.sy, L.sz, - L.sz };
  int nsteps = 0, shell_steps = 0;
  for (size_t q = 0; q < ncell; q++) {
    const int c = seq[q];
    int st = 0;
    for (int d = 0; d < 2*dim; d++) {
      const int sn = step[c + off[d]];
      if (sn > st) st = sn;
    }
    step[c] = st + 1;
    if (st + 1 > nsteps) nsteps = st + 1;
    if (q < nshell && st + 1 > shell_steps) shell_steps = st + 1;
  }
  MpiOrderPlan * P = new MpiOrderPlan;
  P->first.assign (nsteps + 1, 0);
  for (size_t q = 0; q < ncell; q++) P->first[step[seq[q]]]++;    /* counts at [1..nsteps] */
  {
    int acc = 0;
    for (int s = 1; s <= nsteps; s++) { int cnt = P->first[s]; P->first[s] = acc; acc += cnt; }
    /* now first[s] = start of step s (1-based); shift to 0-based */
    for (int s = 0; s < nsteps; s++) P->first[s] = P->first[s + 1];
    P->first[nsteps] = acc;
  }
  std::vector<int> sorted (ncell), fill (P->first.begin (), P->first.end () - 1);
  for (size_t q = 0; q < ncell; q++) {
    const int c = seq[q];
    sorted[fill[step[c] - 1]++] = c;
  }
  P->shell_steps = shell_steps;
  hipError_t e = hipMalloc ((void **) &P->cells, ncell*sizeof (int));
  if (e == hipSuccess)
    e = hipMemcpy (P->cells, sorted.data (), ncell*sizeof (int), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (P->cells) (void) hipFree (P->cells);
    delete P;
    return hip_fail (e, "sweep order", __FILE__, __LINE__);
  }
  dom->mpi_plan[level] = P;
  *out = P;
  return GFSHIP_OK;
}

// one sweep in that order; after the steps that hold the cells along the MPI sides `after_shell'
// is called (the caller starts the halo exchange there)
int launch_relax_mpi_first (gfship_domain * dom, unsigned dimension, int level, double omega,
			    double * u, const double * rhs, const double * dia, const RelaxOp * op,
			    int (* after_shell) (void *), void * ctx)
{
  MpiOrderPlan * P;
  int r = mpi_order_plan (dom, level, &P);
  if (r) return r;
  const int kind = op ? op->kind : 0;
  const double w = op ? op->w : 1., h2 = op ? op->h2 : 1.;
  const Layout & L = dom->lay[level];
  const int nsteps = (int) P->first.size () - 1;
  for (int s = 0; s < nsteps; s++) {
    const int count = P->first[s + 1] - P->first[s];
    const int * cells = P->cells + P->first[s];
    const int block = 256, grid = (count + block - 1)/block;
#define LS_LAUNCH(D, O) hipLaunchKernelGGL ((relax_listed_kernel<D, O>), dim3 (grid), dim3 (block), 0, \
					    dom->stream, cells, count, L.sy, L.sz, dimension, omega, \
					    w, h2, u, rhs, dia)
    if (dom->dim == 3) { if (kind) LS_LAUNCH (3, 1); else LS_LAUNCH (3, 0); }
    else               { if (kind) LS_LAUNCH (2, 1); else LS_LAUNCH (2, 0); }
#undef LS_LAUNCH
    if (s + 1 == P->shell_steps && after_shell && (r = (* after_shell) (ctx)))
      return r;
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

} // namespace gfship
