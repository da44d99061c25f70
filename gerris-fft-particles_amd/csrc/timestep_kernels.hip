// timestep_kernels.hip -- CDNA4 kernels of the projection and Godunov advection path
// (K8-K19 of SURVEY.md 2.5) on the leaf level of a uniform box.
//
// Face state.  The reference keeps two copies of every face quantity, one in each adjacent
// cell (GfsStateVector.f[d].un / .v, src/fluid.h:39-52).  Here:
//   un[c]   one array per component: entry of cell a = normal velocity on the face on the
//           + side of a along c (so a.f[2c].un = un[c][a], a.f[2c+1].un = un[c][a - off_c]);
//           valid for 0 <= coordinate c <= n (index 0 is the ghost cell's + face).
//   fv[d]   six arrays: fv[d][a] = a.f[d].v (Godunov face value of the advected variable),
//           ghosts filled by the face-BC kernel.
// On a uniform grid both copies of the reference are always equal, so this is not a change
// of algorithm, only of layout.  All formulas keep the reference's operand order; where the
// reference scatters (+=) into both cells of a face in traversal order, the kernels gather per
// cell in that same order (see flux_update_kernel).
#include "gfship_internal.hpp"
#include <cfloat>

namespace gfship {

#define CELL_PROLOGUE(L)						\
  int i = blockIdx.x*blockDim.x + threadIdx.x + 1;			\
  int j = blockIdx.y + 1;						\
  int k = (L).dim == 3 ? blockIdx.z + 1 : 0;				\
  if (i > (L).n) return;						\
  long c = (L).idx (i, j, k)

// cells 0..n in every direction (faces on the + side of the low ghost layer included)
#define EXT_PROLOGUE(L)							\
  int i = blockIdx.x*blockDim.x + threadIdx.x;				\
  int j = blockIdx.y;							\
  int k = (L).dim == 3 ? blockIdx.z : 0;				\
  if (i > (L).n) return;						\
  long c = (L).idx (i, j, k)

static inline void cell_grid (const Layout & L, dim3 * grid, dim3 * block)
{
  int b = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
  *block = dim3 (b);
  *grid = dim3 ((L.n + b - 1)/b, L.n, L.dim == 3 ? L.n : 1);
}

static inline void ext_grid (const Layout & L, dim3 * grid, dim3 * block)
{
  int b = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
  *block = dim3 (b);
  *grid = dim3 ((L.n + 1 + b - 1)/b, L.n + 1, L.dim == 3 ? L.n + 1 : 1);
}

struct Ptr3  { double * p[3]; };
struct CPtr3 { const double * p[3]; };
struct Ptr6  { double * p[6]; };
struct CPtr6 { const double * p[6]; };
struct Visc3 { double d[3]; double g[3]; };   // diffusion coefficients and GfsSource intensities of U, V, W

// gfs_face_interpolated_value, src/fluid.c:2186-2198, same-level neighbour (x1 = 1.)
__device__ __forceinline__ double face_interp (double v0, double v1)
{
  double x1 = 1.;
  return ((x1 - 0.5)*v0 + 0.5*v1)/x1;
}

// is (i,j,k) a valid "+ face of cell" location for component cc?
template <int DIM>
__device__ __forceinline__ bool face_valid (int n, int cc, int i, int j, int k)
{
  int q[3] = { i, j, k };
#pragma unroll
  for (int a = 0; a < DIM; a++)
    if (a != cc && (q[a] < 1 || q[a] > n))
      return false;
  return true; /* coordinate cc is in [0,n] by construction of the launch */
}

// K12: gfs_face_interpolated_normal_velocity, src/advection.c:549-573 (after the reset :575-587)
template <int DIM>
__global__ void __launch_bounds__(256)
face_interp_un_kernel (Layout L, CPtr3 u, Ptr3 un)
{
  EXT_PROLOGUE (L);
  const long off[3] = { 1, L.sy, L.sz };
#pragma unroll
  for (int cc = 0; cc < DIM; cc++)
    if (face_valid<DIM> (L.n, cc, i, j, k))
      un.p[cc][c] = face_interp (u.p[cc][c], u.p[cc][c + off[cc]]);
}

// K9: gfs_normal_divergence (src/fluid.c:2310-2324) then scale_divergence (src/timestep.c:181-187)
template <int DIM>
__global__ void __launch_bounds__(256)
divergence_kernel (Layout L, CPtr3 un, double * __restrict__ div, double dt)
{
  CELL_PROLOGUE (L);
  const long off[3] = { 1, L.sy, L.sz };
  double h = 1./L.n;
  double d_ = 0.;
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    d_ += 1.*un.p[cc][c]*1.;
    d_ += -1.*un.p[cc][c - off[cc]]*1.;
  }
  double v = d_*h;
  div[c] = v/dt;
}

// K10: correct_normal_velocity (src/timestep.c:118-144), unit weights:
//   dp = (w*p[nb] - w*p[cell])/h; dp /= face_fraction (= 1.); un -= dp*dt
template <int DIM>
__global__ void __launch_bounds__(256)
correct_un_kernel (Layout L, const double * __restrict__ p, Ptr3 un, double dt)
{
  EXT_PROLOGUE (L);
  const long off[3] = { 1, L.sy, L.sz };
  double h = 1./L.n;
#pragma unroll
  for (int cc = 0; cc < DIM; cc++)
    if (face_valid<DIM> (L.n, cc, i, j, k)) {
      double dp = (1.*p[c + off[cc]] - 1.*p[c])/h;
      dp /= 1.;
      un.p[cc][c] -= dp*dt;
    }
}

// K10 + K11a: centred gradient: g[c] = (dp(- face)*1. + dp(+ face)*1.)/2.
// (accumulation in correct_normal_velocity, halving in scale_cell_gradients src/timestep.c:60-87;
// two addends only, so the traversal order of the two faces does not matter)
template <int DIM>
__global__ void __launch_bounds__(256)
centered_gradient_kernel (Layout L, const double * __restrict__ p, Ptr3 g)
{
  CELL_PROLOGUE (L);
  const long off[3] = { 1, L.sy, L.sz };
  double h = 1./L.n;
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    double dpm = (1.*p[c] - 1.*p[c - off[cc]])/h;
    dpm /= 1.;
    double dpp = (1.*p[c + off[cc]] - 1.*p[c])/h;
    dpp /= 1.;
    double v = 0.;
    v += dpm*1.;
    v += dpp*1.;
    g.p[cc][c] = v/2.;
  }
}

// K11b: correct (src/timestep.c:486-496): u[c] -= g[c]*dt
template <int DIM>
__global__ void __launch_bounds__(256)
correct_centered_kernel (Layout L, Ptr3 u, CPtr3 g, double dt)
{
  CELL_PROLOGUE (L);
#pragma unroll
  for (int cc = 0; cc < DIM; cc++)
    u.p[cc][c] -= g.p[cc][c]*dt;
}

// K11b together with the first level of gfs_cell_coarse_init (src/adaptive.c:43-58) of the corrected
// velocities: one thread per cell of the level below the leaves corrects its eight children and
// leaves their average (children in child-id order: bit 0 -> +x, bit 1 -> -y, bit 2 -> -z) in the
// parent -- the pass over U, V, W that coarse_init_kernel would make right afterwards is saved.
__global__ void __launch_bounds__(256)
correct_centered_coarse_kernel (Layout L, Layout Lc, Ptr3 u, CPtr3 g, double dt, Ptr3 uc)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  int i = blockIdx.x*blockDim.x + threadIdx.x + 1;
  int j = blockIdx.y + 1, k = blockIdx.z + 1;
  if (i > Lc.n) return;
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    double x[2][2][2];        // [dz: ck = 2k, 2k - 1][dy: cj = 2j, 2j - 1][dx: ci = 2i - 1, 2i]
#pragma unroll
    for (int dz = 0; dz < 2; dz++)
#pragma unroll
      for (int dy = 0; dy < 2; dy++) {
	const long c = L.idx (2*i - 1, 2*j - dy, 2*k - dz);       /* even index: 16-byte aligned */
	const d2 a = *(const d2 *) (u.p[cc] + c), b = *(const d2 *) (g.p[cc] + c);
	d2 r;
	r.x = a.x - b.x*dt; r.y = a.y - b.y*dt;
	*(d2 *) (u.p[cc] + c) = r;
	x[dz][dy][0] = r.x; x[dz][dy][1] = r.y;
      }
    double val = 0., sa = 0.;
#pragma unroll
    for (int id = 0; id < 8; id++) {
      val += x[(id >> 2) & 1][(id >> 1) & 1][id & 1]*1.;
      sa += 1.;
    }
    uc.p[cc][Lc.idx (i, j, k)] = val/sa;
  }
}

// K10 + K11 in one pass over the pressure: correct_normal_velocity on every face, the centred
// gradient of every cell (both kernels above, same expressions: each thread recomputes the
// gradient of its - faces instead of reading it back), optionally K11b on the centred velocities
// (approximate projection) and then the largest |un|, |u| for the CFL condition: (h/|x|)^2 is
// monotonically non-increasing in |x|, roundings included, so the minimum of minimum_mac_cfl /
// minimum_cfl (src/domain.c:2824-2890) is that expression of the maximum.
// h = 1/n is a power of two: /h is written as an exact scaling by n.
__device__ __forceinline__ double wave_max_d (double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v = fmax (v, __shfl_down (v, o, 64));
  return v;
}

// Index space: one thread per (i,j,k) with i in 1..n, j and k in 0..n; the thread of i = 1 also
// takes the column i = 0 (faces on the low x side), so that rows map to whole 256-thread blocks.
template <int DIM, bool FUSE_U>
__global__ void __launch_bounds__(256)
project_correct_kernel (Layout L, const double * __restrict__ p, Ptr3 un, Ptr3 g, Ptr3 u,
			double dt, double * __restrict__ partial_max)
{
  const int i1 = blockIdx.x*blockDim.x + threadIdx.x + 1;
  const int j = blockIdx.y;
  const int k = DIM == 3 ? blockIdx.z : 0;
  const int n = L.n;
  const double rn = (double) n;
  const long off[3] = { 1, L.sy, L.sz };
  double mx = 0.;
  auto body = [&] (int i) {
    const long c = L.idx (i, j, k);
    const bool interior = i >= 1 && j >= 1 && (DIM == 2 || k >= 1);
#pragma unroll
    for (int cc = 0; cc < DIM; cc++) {
      const bool valid = face_valid<DIM> (n, cc, i, j, k);
      if (valid || interior) {
	double dpp = (1.*p[c + off[cc]] - 1.*p[c])*rn;
	dpp /= 1.;
	if (valid) {
	  double w = un.p[cc][c];
	  w -= dpp*dt;
	  un.p[cc][c] = w;
	  mx = fmax (mx, fabs (w));
	}
	if (interior) {
	  double dpm = (1.*p[c] - 1.*p[c - off[cc]])*rn;
	  dpm /= 1.;
	  double v = 0.;
	  v += dpm*1.;
	  v += dpp*1.;
	  double gg = v/2.;
	  g.p[cc][c] = gg;
	  if (FUSE_U) {
	    double w = u.p[cc][c];
	    w -= gg*dt;
	    u.p[cc][c] = w;
	    mx = fmax (mx, fabs (1.*w));
	  }
	}
      }
    }
  };
  if (i1 <= n) {
    body (i1);
    if (i1 == 1)
      body (0);
  }
  if (partial_max) {
    __shared__ double sh[4];
    mx = wave_max_d (mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads ();
    if (threadIdx.x == 0) {
      double r = sh[0];
      for (int q = 1; q < (int) (blockDim.x >> 6); q++) r = fmax (r, sh[q]);
      // non-negative doubles order like their bit patterns: one atomic per block into 1024 slots
      if (r > 0.)
	atomicMax ((unsigned long long *) partial_max +
		   ((blockIdx.x + gridDim.x*(blockIdx.y + (size_t) gridDim.y*blockIdx.z)) & 1023),
		   (unsigned long long) __double_as_longlong (r));
    }
  }
}

// face_interp_div_kernel<3, false> (the divergence of the interpolated MAC velocities, nothing stored on the
// faces) with two cells of a row per thread and 16-byte accesses: 9 loads and 1 store per two cells instead of
// 18 and 2; only the interior rows have anything to do
__global__ void __launch_bounds__(128)
face_interp_div2_kernel (Layout L, CPtr3 u, double * __restrict__ div, double dt)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int i1 = 2*(blockIdx.x*blockDim.x + threadIdx.x) + 1;
  const int j = blockIdx.y + 1, k = blockIdx.z + 1;
  if (i1 > L.n) return;
  const double h = 1./L.n;
  const long c = L.idx (i1, j, k);
  const long off[3] = { 1, L.sy, L.sz };
  d2 D = { 0., 0. };
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    const d2 U = *(const d2 *) (u.p[cc] + c);
    d2 Un, Um;
    if (cc == 0) { Un.x = U.y; Un.y = u.p[0][c + 2]; Um.x = u.p[0][c - 1]; Um.y = U.x; }
    else { Un = *(const d2 *) (u.p[cc] + c + off[cc]); Um = *(const d2 *) (u.p[cc] + c - off[cc]); }
    {
      const double unp = face_interp (U.x, Un.x), unm = face_interp (Um.x, U.x);
      D.x += 1.*unp*1.;
      D.x += -1.*unm*1.;
    }
    {
      const double unp = face_interp (U.y, Un.y), unm = face_interp (Um.y, U.y);
      D.y += 1.*unp*1.;
      D.y += -1.*unm*1.;
    }
  }
  d2 out;
  { const double v = D.x*h; out.x = v/dt; }
  { const double v = D.y*h; out.y = v/dt; }
  *(d2 *) (div + c) = out;
}

// project_correct_kernel<3, false> (the MAC projection: correct_normal_velocity + the centred gradient) with two
// cells of a row per thread and 16-byte accesses, as project_correct_lazy2_kernel below
__global__ void __launch_bounds__(128)
project_correct2_kernel (Layout L, const double * __restrict__ p, Ptr3 un, Ptr3 g, double dt)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int i1 = 2*(blockIdx.x*blockDim.x + threadIdx.x) + 1;
  const int j = blockIdx.y, k = blockIdx.z;
  const int n = L.n;
  const double rn = (double) n;
  const long off[3] = { 1, L.sy, L.sz };
  auto body = [&] (int i) {
    const long c = L.idx (i, j, k);
    const bool interior = i >= 1 && j >= 1 && k >= 1;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      const bool valid = face_valid<3> (n, cc, i, j, k);
      if (valid || interior) {
	double dpp = (1.*p[c + off[cc]] - 1.*p[c])*rn;
	dpp /= 1.;
	if (valid) {
	  double w = un.p[cc][c];
	  w -= dpp*dt;
	  un.p[cc][c] = w;
	}
	if (interior) {
	  double dpm = (1.*p[c] - 1.*p[c - off[cc]])*rn;
	  dpm /= 1.;
	  double v = 0.;
	  v += dpm*1.;
	  v += dpp*1.;
	  g.p[cc][c] = v/2.;
	}
      }
    }
  };
  if (i1 > n) return;
  if (j >= 1 && k >= 1) {
    const long c = L.idx (i1, j, k);
    const d2 P = *(const d2 *) (p + c);
    const double pl = p[c - 1], pr = p[c + 2];
    const d2 Pn[3] = { { P.y, pr }, *(const d2 *) (p + c + L.sy), *(const d2 *) (p + c + L.sz) };
    const d2 Pm[3] = { { pl, P.x }, *(const d2 *) (p + c - L.sy), *(const d2 *) (p + c - L.sz) };
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      d2 W = *(const d2 *) (un.p[cc] + c), G;
#pragma unroll
      for (int e = 0; e < 2; e++) {
	const double pc = e ? P.y : P.x, pn = e ? Pn[cc].y : Pn[cc].x, pm = e ? Pm[cc].y : Pm[cc].x;
	double dpp = (1.*pn - 1.*pc)*rn;
	dpp /= 1.;
	double w = e ? W.y : W.x;
	w -= dpp*dt;
	double dpm = (1.*pc - 1.*pm)*rn;
	dpm /= 1.;
	double v = 0.;
	v += dpm*1.;
	v += dpp*1.;
	if (e) { W.y = w; G.y = v/2.; } else { W.x = w; G.x = v/2.; }
      }
      *(d2 *) (un.p[cc] + c) = W;
      *(d2 *) (g.p[cc] + c) = G;
    }
  }
  else {
    body (i1);
    if (i1 + 1 <= n) body (i1 + 1);
  }
  if (i1 == 1)
    body (0);
}

// project_correct_lazy_kernel<3> with two cells of a row per thread: the rows of a level start 16-byte aligned
// at i = 1 (Layout), so p, its four neighbour rows, the three velocity components, their + neighbours along y
// and z and the six results move as 16-byte accesses (19 memory instructions per two cells instead of 38: the
// 8-byte version is bound by the number of requests, not by bytes).  Same expressions per cell.  Rows of the
// ghost planes (j = 0 or k = 0) and the column i = 0 keep the one-cell body.
__global__ void __launch_bounds__(128)
project_correct_lazy2_kernel (Layout L, const double * __restrict__ p, CPtr3 u, Ptr3 g, Ptr3 uo,
			      double dt, double * __restrict__ partial_max)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int i1 = 2*(blockIdx.x*blockDim.x + threadIdx.x) + 1;      /* odd: the pair (i1, i1 + 1) */
  const int j = blockIdx.y, k = blockIdx.z;
  const int n = L.n;
  const double rn = (double) n;
  const long off[3] = { 1, L.sy, L.sz };
  double mx = 0.;
  auto body = [&] (int i) {
    const long c = L.idx (i, j, k);
    const bool interior = i >= 1 && j >= 1 && k >= 1;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      const bool valid = face_valid<3> (n, cc, i, j, k);
      if (valid || interior) {
	double dpp = (1.*p[c + off[cc]] - 1.*p[c])*rn;
	dpp /= 1.;
	const double uc = u.p[cc][c];
	if (valid) {
	  double w = face_interp (uc, u.p[cc][c + off[cc]]);
	  w -= dpp*dt;
	  mx = fmax (mx, fabs (w));
	}
	if (interior) {
	  double dpm = (1.*p[c] - 1.*p[c - off[cc]])*rn;
	  dpm /= 1.;
	  double v = 0.;
	  v += dpm*1.;
	  v += dpp*1.;
	  double gg = v/2.;
	  g.p[cc][c] = gg;
	  double w = uc;
	  w -= gg*dt;
	  uo.p[cc][c] = w;
	  mx = fmax (mx, fabs (1.*w));
	}
      }
    }
  };
  if (i1 <= n) {
    if (j >= 1 && k >= 1) {
      // every face of the two cells is valid and both are interior cells
      const long c = L.idx (i1, j, k);
      const d2 P = *(const d2 *) (p + c);
      const double pl = p[c - 1], pr = p[c + 2];
      const d2 Pn[3] = { { P.y, pr }, *(const d2 *) (p + c + L.sy), *(const d2 *) (p + c + L.sz) };
      const d2 Pm[3] = { { pl, P.x }, *(const d2 *) (p + c - L.sy), *(const d2 *) (p + c - L.sz) };
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {
	const d2 U = *(const d2 *) (u.p[cc] + c);
	d2 Un;
	if (cc == 0) { Un.x = U.y; Un.y = u.p[0][c + 2]; }
	else Un = *(const d2 *) (u.p[cc] + c + off[cc]);
	d2 G, W;
#pragma unroll
	for (int e = 0; e < 2; e++) {
	  const double pc = e ? P.y : P.x, pn = e ? Pn[cc].y : Pn[cc].x, pm = e ? Pm[cc].y : Pm[cc].x;
	  const double uc = e ? U.y : U.x, un_ = e ? Un.y : Un.x;
	  double dpp = (1.*pn - 1.*pc)*rn;
	  dpp /= 1.;
	  double w = face_interp (uc, un_);
	  w -= dpp*dt;
	  mx = fmax (mx, fabs (w));
	  double dpm = (1.*pc - 1.*pm)*rn;
	  dpm /= 1.;
	  double v = 0.;
	  v += dpm*1.;
	  v += dpp*1.;
	  const double gg = v/2.;
	  double w2 = uc;
	  w2 -= gg*dt;
	  mx = fmax (mx, fabs (1.*w2));
	  if (e) { G.y = gg; W.y = w2; } else { G.x = gg; W.x = w2; }
	}
	*(d2 *) (g.p[cc] + c) = G;
	*(d2 *) (uo.p[cc] + c) = W;
      }
    }
    else {
      body (i1);
      if (i1 + 1 <= n) body (i1 + 1);
    }
    if (i1 == 1)
      body (0);
  }
  if (partial_max) {
    __shared__ double sh[4];
    mx = wave_max_d (mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads ();
    if (threadIdx.x == 0) {
      double r = sh[0];
      for (int q = 1; q < (int) (blockDim.x >> 6); q++) r = fmax (r, sh[q]);
      if (r > 0.)
	atomicMax ((unsigned long long *) partial_max +
		   ((blockIdx.x + gridDim.x*(blockIdx.y + (size_t) gridDim.y*blockIdx.z)) & 1023),
		   (unsigned long long) __double_as_longlong (r));
    }
  }
}

// The same pass with the face weights f[d].v of gfs_poisson_coefficients with a GfsFunction alpha
// (GfsPhysicalParams { alpha = ... }: variable density): correct_normal_velocity, src/timestep.c:118-144,
// takes gfs_face_weighted_gradient (src/fluid.c:858-864: g.a = w, g.b = w*p_nb) with the weight of the
// cell the face traversal visits the face FROM -- the cell on the low side (its direction 2 c), except
// for the faces on the low side of the box, visited from the border cell in its direction 2 c + 1
// (src/ftt.c:2152-2215).  dp = (w p_nb - w p_cell)/h either way (the change of sign of an odd direction
// is exact).  un == nullptr: gfs_update_gradients (dt = 0: the face velocities stay as they are).
struct W6t { const double * p[6]; };

template <int DIM>
__global__ void __launch_bounds__(256)
project_correct_weighted_kernel (Layout L, const double * __restrict__ p, Ptr3 un, Ptr3 g, Ptr3 u,
				 double dt, W6t wf)
{
  const int i1 = blockIdx.x*blockDim.x + threadIdx.x + 1;
  const int j = blockIdx.y;
  const int k = DIM == 3 ? blockIdx.z : 0;
  const int n = L.n;
  const double rn = (double) n;
  const long off[3] = { 1, L.sy, L.sz };
  auto body = [&] (int i) {
    const long c = L.idx (i, j, k);
    const int q[3] = { i, j, k };
    const bool interior = i >= 1 && j >= 1 && (DIM == 2 || k >= 1);
#pragma unroll
    for (int cc = 0; cc < DIM; cc++) {
      const bool valid = face_valid<DIM> (n, cc, i, j, k);
      if (valid || interior) {
	// + face of c: visited from c unless c is the ghost cell in front of the box
	const double wp = q[cc] >= 1 ? wf.p[2*cc][c] : wf.p[2*cc + 1][c + off[cc]];
	double dpp = (wp*p[c + off[cc]] - wp*p[c])*rn;
	dpp /= 1.;
	if (valid && un.p[cc]) {
	  double w = un.p[cc][c];
	  w -= dpp*dt;
	  un.p[cc][c] = w;
	}
	if (interior) {
	  // - face of c: visited from the cell below unless that is the ghost cell
	  const double wm = q[cc] >= 2 ? wf.p[2*cc][c - off[cc]] : wf.p[2*cc + 1][c];
	  double dpm = (wm*p[c] - wm*p[c - off[cc]])*rn;
	  dpm /= 1.;
	  double v = 0.;
	  v += dpm*1.;
	  v += dpp*1.;
	  double gg = v/2.;
	  g.p[cc][c] = gg;
	  if (u.p[cc]) {
	    double w = u.p[cc][c];
	    w -= gg*dt;
	    u.p[cc][c] = w;
	  }
	}
      }
    }
  };
  if (i1 <= n) {
    body (i1);
    if (i1 == 1)
      body (0);
  }
}

// The update of the approximate projection when nobody reads the corrected MAC velocities before the
// next predictor overwrites them (the loop body of simulation_run without tracers): the face
// velocities are not stored at all -- un = face_interp (u) - dp dt is formed in registers for the
// maximum of the CFL condition only -- and the centred velocities are corrected out of place (uo:
// the faces read the uncorrected neighbours), the caller swaps the storage.  80 B per cell instead of
// 128 (and 32 instead of 56 in the interpolation + divergence pass before it, which no longer writes
// un).  Should somebody ask for the MAC velocities after all, they are rebuilt from the uncorrected
// velocities kept in the swapped-out storage and p by face_interp_un_kernel + correct_un_kernel: the
// same expressions, the same bits (materialize_un, simulation.hip).
template <int DIM>
__global__ void __launch_bounds__(256)
project_correct_lazy_kernel (Layout L, const double * __restrict__ p, CPtr3 u, Ptr3 g, Ptr3 uo,
			     double dt, double * __restrict__ partial_max)
{
  const int i1 = blockIdx.x*blockDim.x + threadIdx.x + 1;
  const int j = blockIdx.y;
  const int k = DIM == 3 ? blockIdx.z : 0;
  const int n = L.n;
  const double rn = (double) n;
  const long off[3] = { 1, L.sy, L.sz };
  double mx = 0.;
  auto body = [&] (int i) {
    const long c = L.idx (i, j, k);
    const bool interior = i >= 1 && j >= 1 && (DIM == 2 || k >= 1);
#pragma unroll
    for (int cc = 0; cc < DIM; cc++) {
      const bool valid = face_valid<DIM> (n, cc, i, j, k);
      if (valid || interior) {
	double dpp = (1.*p[c + off[cc]] - 1.*p[c])*rn;
	dpp /= 1.;
	const double uc = u.p[cc][c];
	if (valid) {
	  double w = face_interp (uc, u.p[cc][c + off[cc]]);
	  w -= dpp*dt;
	  mx = fmax (mx, fabs (w));
	}
	if (interior) {
	  double dpm = (1.*p[c] - 1.*p[c - off[cc]])*rn;
	  dpm /= 1.;
	  double v = 0.;
	  v += dpm*1.;
	  v += dpp*1.;
	  double gg = v/2.;
	  g.p[cc][c] = gg;
	  double w = uc;
	  w -= gg*dt;
	  uo.p[cc][c] = w;
	  mx = fmax (mx, fabs (1.*w));
	}
      }
    }
  };
  if (i1 <= n) {
    body (i1);
    if (i1 == 1)
      body (0);
  }
  if (partial_max) {
    __shared__ double sh[4];
    mx = wave_max_d (mx);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads ();
    if (threadIdx.x == 0) {
      double r = sh[0];
      for (int q = 1; q < (int) (blockDim.x >> 6); q++) r = fmax (r, sh[q]);
      if (r > 0.)
	atomicMax ((unsigned long long *) partial_max +
		   ((blockIdx.x + gridDim.x*(blockIdx.y + (size_t) gridDim.y*blockIdx.z)) & 1023),
		   (unsigned long long) __double_as_longlong (r));
    }
  }
}

// cfl^2 = (h/max)^2 from the 1024 slots, which are cleared for the next use
__global__ void __launch_bounds__(256)
cfl_from_max_kernel (double * __restrict__ partial_max, int nblocks, double length,
		     double * __restrict__ out)
{
  double m = 0.;
  for (int q = threadIdx.x; q < nblocks; q += blockDim.x) {
    m = fmax (m, partial_max[q]);
    partial_max[q] = 0.;
  }
  __shared__ double sh[4];
  m = wave_max_d (m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads ();
  if (threadIdx.x == 0) {
    m = fmax (fmax (sh[0], sh[1]), fmax (sh[2], sh[3]));
    double r = DBL_MAX;
    if (m != 0.) {
      double cflu = length/fabs (m);
      r = cflu*cflu;
    }
    out[0] = r;
  }
}

// K12 + K9 in one pass over the centred velocities (approximate projection): the MAC velocities
// and their divergence scaled by 1/dt; the - face of a cell is recomputed, not read back
// WRITE_UN = false: only the divergence is wanted (the approximate projection of simulation_run, whose
// corrected MAC velocities nobody reads: project_correct_lazy_kernel)
template <int DIM, bool WRITE_UN>
__global__ void __launch_bounds__(256)
face_interp_div_kernel (Layout L, CPtr3 u, Ptr3 un, double * __restrict__ div, double dt)
{
  const int i1 = blockIdx.x*blockDim.x + threadIdx.x + 1;
  const int j = blockIdx.y;
  const int k = DIM == 3 ? blockIdx.z : 0;
  if (i1 > L.n) return;
  const long off[3] = { 1, L.sy, L.sz };
  const double h = 1./L.n;
  auto body = [&] (int i) {
    const long c = L.idx (i, j, k);
    const bool interior = i >= 1 && j >= 1 && (DIM == 2 || k >= 1);
    double d_ = 0.;
#pragma unroll
    for (int cc = 0; cc < DIM; cc++) {
      const bool valid = face_valid<DIM> (L.n, cc, i, j, k);
      double unp = 0.;
      if (valid || interior)
	unp = face_interp (u.p[cc][c], u.p[cc][c + off[cc]]);
      if (WRITE_UN && valid)
	un.p[cc][c] = unp;
      if (interior) {
	double unm = face_interp (u.p[cc][c - off[cc]], u.p[cc][c]);
	d_ += 1.*unp*1.;
	d_ += -1.*unm*1.;
      }
    }
    if (interior) {
      double v = d_*h;
      div[c] = v/dt;
    }
  };
  body (i1);
  if (i1 == 1)
    body (0);
}

// ---------------------------------------------------------------------------------------------
// K13: gfs_cell_advected_face_values, src/advection.c:58-99
// ---------------------------------------------------------------------------------------------
#define GMIN(a, b) (((a) < (b)) ? (a) : (b))
#define GMAX(a, b) (((a) > (b)) ? (a) : (b))
#define GABS(a)    (((a) < 0) ? -(a) : (a))

// gfs_center_gradient (src/fluid.c:434-457), both neighbours at the same level
__device__ __forceinline__ double center_gradient (double v0, double v1, double v2)
{
  double x1 = 1., x2 = 1.;
  return (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
}

// gfs_center_van_leer_gradient (src/fluid.c:522-561)
__device__ __forceinline__ double van_leer_gradient (double v0, double v1, double v2)
{
  double x1 = 1., x2 = 1.;
  double s1 = 2.*(v0 - v1);
  double s2 = 2.*(v2 - v0);
  if (s1*s2 <= 0.)
    return 0.;
  double s0 = (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
  if (GABS (s2) < GABS (s1))
    s1 = s2;
  if (GABS (s0) < GABS (s1))
    return s0;
  return s1;
}

// gfs_center_minmod_gradient / superbee / sweby (src/fluid.c:563-690): center_limited_gradient with
// generic_limiter (r, beta), beta = 1, 2, 1.5; both neighbours at the same level (x1 = x2 = 1.)
__device__ __forceinline__ double limited_gradient (double v0, double v1, double v2, double beta)
{
  const double x1 = 1., x2 = 1.;
  if (v0 == v1)
    return 0.;
  const double r = (v2 - v0)*x1/((v0 - v1)*x2);
  double l1 = GMIN (r, beta), l2 = GMIN (beta*r, 1.);
  l1 = GMAX (0., l1);
  const double lim = GMAX (l1, l2);
  return lim*(v0 - v1)/x1;
}

// GfsAdvectionParams.gradient / GfsVariableTracer { gradient = }: 0 gfs_center_gradient, 1 van Leer,
// 2 minmod, 3 superbee, 4 sweby (the kernels with the gradient as a template parameter know 0 and 1:
// the limiters of kinds 2 .. 4 run on the general path)
__device__ __forceinline__ double cell_gradient (int kind, double v0, double v1, double v2)
{
  switch (kind) {
  case 0: return center_gradient (v0, v1, v2);
  case 1: return van_leer_gradient (v0, v1, v2);
  case 2: return limited_gradient (v0, v1, v2, 1.);
  case 3: return limited_gradient (v0, v1, v2, 2.);
  default: return limited_gradient (v0, v1, v2, 1.5);
  }
}

// source_diffusion_value, src/source.c:1105-1144: the explicit diffusion term of an implicit
// GfsSourceDiffusion with constant coefficient D (gfs_face_gradient at the cell's level:
// e.a = 1., e.b = neighbour value), used as MAC source of the predictor and in the CFL scale
template <int DIM>
__device__ __forceinline__ double source_diffusion_value (const double * __restrict__ v, long c,
							  const long * off, double D, int n)
{
  /* h = 1/n is a power of two: the division by h*h is an exact scaling by n*n */
  const double rh2 = (double) n*(double) n;
  double ga = 0., gb = 0.;
  const double v0 = v[c];
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    ga += D*1.; gb += D*v[c + off[cc]];
    ga += D*1.; gb += D*v[c - off[cc]];
  }
  return 1.*(gb - ga*v0)*rh2;
}

template <int DIM>
__global__ void __launch_bounds__(256)
advected_face_values_kernel (Layout L, const double * __restrict__ v, CPtr3 u, CPtr3 un,
			     double dt, int use_centered_velocity, int gradient, Ptr6 fv, int cmask,
			     double visc, double gsrc)
{
  CELL_PROLOGUE (L);
  const long off[3] = { 1, L.sy, L.sz };
  const double size = 1./L.n;
  const double v0 = v[c];
  // tangential / normal velocities
  double vel[3];
#pragma unroll
  for (int cc = 0; cc < DIM; cc++)
    vel[cc] = use_centered_velocity ? u.p[cc][c] : 0.;
  // transverse_term, src/advection.c:27-47
  double tt[3];
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    double vtan = use_centered_velocity ? vel[cc] : (un.p[cc][c] + un.p[cc][c - off[cc]])/2.;
    long nb = vtan > 0. ? c - off[cc] : c + off[cc];
    double g = v[nb] - 1.*v0;         /* gfs_face_gradient: g.b - g.a*v0, src/fluid.c:801-805 */
    if (vtan > 0.) g = - g;
    tt[cc] = dt*vtan*g/(2.*size);
  }
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    double unorm = use_centered_velocity ?
      dt*vel[cc]/size :
      dt*(un.p[cc][c] + un.p[cc][c - off[cc]])/(2.*size);
    double v1 = v[c - off[cc]], v2 = v[c + off[cc]];
    double g = cell_gradient (gradient, v0, v1, v2);
    double vl = v0 + GMIN ((1. - unorm)/2., 0.5)*g;
    double vr = v0 + GMAX ((- 1. - unorm)/2., -0.5)*g;
    /* gfs_variable_mac_source, src/source.c:38-59 */
    double msrc = 0.;
    if (visc != 0.)
      msrc = 0. + source_diffusion_value<DIM> (v, c, off, visc, L.n);
    msrc += gsrc;               /* the intensity of a GfsSource (source_value, src/source.c:398-403) */
    double src = dt*msrc/2.;
    double dv;
    if (DIM == 2)
      dv = tt[(cc + 1) % 2];
    else {
      /* orthogonal[c] = {Y,Z}, {X,Z}, {X,Y} */
      int o0 = cc == 0 ? 1 : 0, o1 = cc == 2 ? 1 : 2;
      dv = tt[o0];
      dv += tt[o1];
    }
    if (cmask & (1 << cc)) {   /* the predictor only uses the faces normal to the component */
      fv.p[2*cc][c]     = vl + src - dv;
      fv.p[2*cc + 1][c] = vr + src - dv;
    }
  }
}

// gfs_domain_face_bc (src/domain.c:1209-1232): ghost face values.
// face_symmetry src/boundary.c:64-74, face_dirichlet :275-279, face_neumann :354-360,
// face_periodic/face_update :1251-1258,1343-1347.  The ghost cell G beyond side sd keeps its
// value in fv[opposite(sd)][G]; symmetry/Dirichlet also overwrite the interior cell's fv[sd].
template <int DIM>
__global__ void __launch_bounds__(256)
face_bc_kernel (Layout L, BcDesc bc, const double * __restrict__ v, Ptr6 fv, int cmask)
{
  const int n = L.n;
  const int nface = DIM == 3 ? n*n : n;
  int f = blockIdx.x*blockDim.x + threadIdx.x;
  int sd = blockIdx.y;
  if (f >= nface) return;
  if (bc.side[sd] == GFSHIP_SIDE_EXTERNAL) return;
  if (!(cmask & (1 << (sd/2)))) return;
  int cc = sd/2, od = sd ^ 1;
  int t1 = f % n + 1, t2 = DIM == 3 ? f / n + 1 : 0;
  int ijk[3] = { 0, 0, 0 };
  int ta = cc == 0 ? 1 : 0, tb = cc == 2 ? 1 : 2;
  ijk[cc] = (sd & 1) ? 1 : n;
  ijk[ta] = t1;
  if (DIM == 3) ijk[tb] = t2;
  long o = cc == 0 ? 1 : cc == 1 ? L.sy : L.sz;
  if (sd & 1) o = - o;
  long cell = L.idx (ijk[0], ijk[1], ijk[2]);
  long G = cell + o;
  if (bc.side[sd] == GFSHIP_SIDE_PERIODIC) {
    fv.p[od][G] = fv.p[od][cell - (long) (n - 1)*o];
    return;
  }
  double val = bc.val[sd] ? bc.val[sd][f] : 0.;
  switch (bc.type[sd]) {
  case GFSHIP_BC_DIRICHLET:
    fv.p[od][G] = fv.p[sd][cell] = val;
    break;
  case GFSHIP_BC_NEUMANN:
    fv.p[od][G] = v[cell] + val*(1./n)/2.;
    break;
  default:
    if (bc.component == cc)
      fv.p[od][G] = fv.p[sd][cell] = 0.;
    else
      fv.p[od][G] = fv.p[sd][cell];
  }
}

// gfs_face_upwinded_value, src/advection.c:267-300, FTT_FINE_FINE: the face on the + side of
// cell a (direction cc); a = left state owner, a + off = right state owner.
__device__ __forceinline__ double upwinded (double un, double fl, double fr)
{
  return un > 0. ? fl : un < 0. ? fr : (fl + fr)/2.;
}

// K14: gfs_face_advected_normal_velocity with GFS_CENTERED_UPWINDING, src/advection.c:513-539
template <int DIM>
__global__ void __launch_bounds__(256)
predict_un_kernel (Layout L, int cc, const double * __restrict__ uc,
		   const double * __restrict__ fvp, const double * __restrict__ fvm,
		   double * __restrict__ unc)
{
  EXT_PROLOGUE (L);
  if (!face_valid<DIM> (L.n, cc, i, j, k)) return;
  long o = cc == 0 ? 1 : cc == 1 ? L.sy : L.sz;
  double s = face_interp (uc[c], uc[c + o]);
  unc[c] = upwinded (s, fvp[c], fvm[c + o]);
}

// ---------------------------------------------------------------------------------------------
// K15 + K16: gfs_face_velocity_advection_flux / gfs_face_advection_flux (src/advection.c:356-435),
// gfs_advection_update (:817-818) and add_pressure_gradient (src/timestep.c:809-812).
//
// The reference zeroes fv on both sides of every face, then scatters -flux / +flux in face
// traversal order (src/ftt.c:2152-2215): all cells in tree pre-order with their d = 0,2,4 faces,
// then the cells along the d = 1,3,5 sides.  A cell C therefore receives, in this order:
//   + F(x, C - ex)                    when its left neighbour was visited (x grows along the order)
//   - F(x, C), - F(y, C), - F(z, C)   at its own visit
//   + F(y, C - ey), + F(z, C - ez)    from the bottom and back neighbours, visited later
//                                     (y and z decrease along the order), bottom first unless
//                                     the back neighbour precedes it in the pre-order
//   + F on the d = 1, 3, 5 box sides  for cells along those sides (second pass)
// and the gather below adds the same bit patterns in the same order.
// With J = n - j, K = n - k the 0-based oriented coordinates of C, the back neighbour
// (J, K + 1) precedes the bottom neighbour (J + 1, K) in the pre-order iff the lowest zero bit of
// J is above the lowest zero bit of K.
// ---------------------------------------------------------------------------------------------
template <int DIM, bool VELOCITY>
__device__ __forceinline__ double face_flux (const Layout & L, int cf, long a, const CPtr3 & un,
					     const CPtr6 & fv, const double * __restrict__ gm,
					     double dt, double h)
{
  long o = cf == 0 ? 1 : cf == 1 ? L.sy : L.sz;
  double unf = un.p[cf][a];
  double upw = upwinded (unf, fv.p[2*cf][a], fv.p[2*cf + 1][a + o]);
  if (VELOCITY) {
    double flux = 1.*unf*dt/h;
    flux *= upw - face_interp (gm[a], gm[a + o])*dt/2.;
    return flux;
  }
  return 1.*unf*dt*upw/h;
}

template <int DIM, bool VELOCITY>
__global__ void __launch_bounds__(256)
flux_update_kernel (Layout L, double * __restrict__ v, CPtr3 un, CPtr6 fv,
		    const double * __restrict__ gm, const double * __restrict__ gc, double dt, double gsrc)
{
  CELL_PROLOGUE (L);
  const int n = L.n;
  const double h = 1./n;
  double acc = 0.;
  if (i > 1)
    acc += face_flux<DIM, VELOCITY> (L, 0, c - 1, un, fv, gm, dt, h);
  acc -= face_flux<DIM, VELOCITY> (L, 0, c, un, fv, gm, dt, h);
  acc -= face_flux<DIM, VELOCITY> (L, 1, c, un, fv, gm, dt, h);
  if (DIM == 3) {
    acc -= face_flux<DIM, VELOCITY> (L, 2, c, un, fv, gm, dt, h);
    unsigned J = n - j, K = n - k;
    bool back_first = __ffs (~J) > __ffs (~K);
    if (back_first) {
      if (k > 1) acc += face_flux<DIM, VELOCITY> (L, 2, c - L.sz, un, fv, gm, dt, h);
      if (j > 1) acc += face_flux<DIM, VELOCITY> (L, 1, c - L.sy, un, fv, gm, dt, h);
    }
    else {
      if (j > 1) acc += face_flux<DIM, VELOCITY> (L, 1, c - L.sy, un, fv, gm, dt, h);
      if (k > 1) acc += face_flux<DIM, VELOCITY> (L, 2, c - L.sz, un, fv, gm, dt, h);
    }
  }
  else if (j > 1)
    acc += face_flux<DIM, VELOCITY> (L, 1, c - L.sy, un, fv, gm, dt, h);
  if (i == 1)
    acc += face_flux<DIM, VELOCITY> (L, 0, c - 1, un, fv, gm, dt, h);
  if (j == 1)
    acc += face_flux<DIM, VELOCITY> (L, 1, c - L.sy, un, fv, gm, dt, h);
  if (DIM == 3 && k == 1)
    acc += face_flux<DIM, VELOCITY> (L, 2, c - L.sz, un, fv, gm, dt, h);
  double val = v[c];
  val += acc/1.;
  if (gc)
    val -= gc[c]*dt;
  if (gsrc != 0.) {             /* gfs_domain_variable_centered_sources, src/source.c:62-108 */
    double sum = 0;
    sum += gsrc;
    val += dt*sum;
  }
  v[c] = val;
}

// ---------------------------------------------------------------------------------------------
// Fused Godunov step on periodic boxes: K13 + face BC + K14 (predictor) and K13 + face BC + K15 +
// K16 (advection) without the six face-value arrays.
//
// The reference stores the two face values of every cell and direction (f[d].v), applies the
// face BC, then visits the faces.  On a periodic box the face value of a ghost cell is the face
// value of its periodic image (face_periodic, src/boundary.c:1251-1258), so every face value a
// face needs can be recomputed on the spot from the cell-centred field: same expression, same
// operand order, same bits as advected_face_values_kernel -- but 56 B/cell of HBM traffic per
// advected component instead of 184 (and 48 instead of 240 for the three predictor passes).
// ---------------------------------------------------------------------------------------------
struct FacePair { double l, r; };   // f[2*d].v ("left state" of the + face) and f[2*d+1].v

// the two face values of direction D of the cell at index c (an interior cell or the periodic
// image of a ghost cell): gfs_cell_advected_face_values (src/advection.c:58-99) for one direction.
// The cell size is a power of two (1/n, n = 2^level), so the reference's divisions by size and
// 2.*size are exact scalings and are written as multiplications by n and n/2: same bits, without
// the 14-instruction division sequences.
// CEN: centred velocities (predictor) or MAC velocities; VL: van Leer or centred gradient; VS: a
// viscous MAC source may be present (all compile-time: the tiled kernels are issue-bound)
template <int DIM, int D, bool CEN, bool VL, bool VS>
__device__ __forceinline__ FacePair face_values_dir (const Layout & L, const double * __restrict__ v,
						     const CPtr3 & u, const CPtr3 & un, int c,
						     double dt, double visc, double gsrc)
{
  const int use_centered_velocity = CEN, gradient = VL;
  // 32-bit cell indices (a level has far fewer than 2^31 doubles): the loads take a uniform base
  // pointer plus a 32-bit offset instead of 64-bit address arithmetic per access
  const int off[3] = { 1, (int) L.sy, (int) L.sz };
  const double rsize = (double) L.n, rsize2 = (double) L.n/2.;   /* 1/size, 1/(2.*size) */
  const double v0 = v[c];
  double tt[3] = { 0., 0., 0. };
#pragma unroll
  for (int cc = 0; cc < DIM; cc++)
    if (cc != D) {
      double vtan = use_centered_velocity ? u.p[cc][c] : (un.p[cc][c] + un.p[cc][c - off[cc]])/2.;
      int nb = vtan > 0. ? c - off[cc] : c + off[cc];
      double g = v[nb] - 1.*v0;
      if (vtan > 0.) g = - g;
      tt[cc] = dt*vtan*g*rsize2;
    }
  double unorm = use_centered_velocity ?
    dt*u.p[D][c]*rsize :
    dt*(un.p[D][c] + un.p[D][c - off[D]])*rsize2;
  double v1 = v[c - off[D]], v2 = v[c + off[D]];
  double g = gradient ? van_leer_gradient (v0, v1, v2) : center_gradient (v0, v1, v2);
  double vl = v0 + GMIN ((1. - unorm)/2., 0.5)*g;
  double vr = v0 + GMAX ((- 1. - unorm)/2., -0.5)*g;
  double msrc = 0.;
  if (VS && visc != 0.) {
    const long off64[3] = { 1, L.sy, L.sz };
    msrc = 0. + source_diffusion_value<DIM> (v, c, off64, visc, L.n);
  }
  if (VS) msrc += gsrc;         /* the intensity of a GfsSource on the variable */
  double src = dt*msrc/2.;
  double dv;
  if (DIM == 2)
    dv = tt[(D + 1) % 2];
  else {
    const int o0 = D == 0 ? 1 : 0, o1 = D == 2 ? 1 : 2;
    dv = tt[o0];
    dv += tt[o1];
  }
  FacePair f;
  f.l = vl + src - dv;
  f.r = vr + src - dv;
  return f;
}

// index of the interior image of cell (i,j,k), each coordinate in [0, n+1]
template <int DIM>
__device__ __forceinline__ int image (const Layout & L, int i, int j, int k)
{
  const int n = L.n;
  i = i < 1 ? i + n : i > n ? i - n : i;
  j = j < 1 ? j + n : j > n ? j - n : j;
  if (DIM == 3) k = k < 1 ? k + n : k > n ? k - n : k;
  return (int) L.idx (i, j, k);
}

// Tiles of GX x GY x GZ cells, one thread per cell.  Phase 1: every thread computes the face
// values of its own cell once and stores them in LDS; the cells just outside the six faces of
// the tile (periodic images at the box sides) are computed by the first threads, one direction
// each.  Phase 2: the faces are visited from LDS.  1.3 face-value evaluations per cell instead of
// the 3 of a per-face recomputation.
#define GX 32
#define GY 4
#define GZ 4
#define GN (GX*GY*GZ)

struct TileIdx {
  int tx, ty, tz, i, j, k;
  __device__ __forceinline__ TileIdx () {
    int tid = threadIdx.x;
    tx = tid % GX; ty = (tid / GX) % GY; tz = tid / (GX*GY);
    i = blockIdx.x*GX + tx + 1; j = blockIdx.y*GY + ty + 1; k = blockIdx.z*GZ + tz + 1;
  }
  __device__ __forceinline__ int own () const { return tx + GX*(ty + GY*tz); }
};

struct GhostFv {
  const double * r[6];     // received states beyond side d; nullptr: the side is not an MPI side
  double * s[6];           // states to send across side d
};

// face position f of a ghost / layer cell of direction c: f = (x_a - 1) + n (x_b - 1) with a < b the
// two directions other than c
__device__ __forceinline__ int face_pos (int c, int n, int i, int j, int k)
{
  return c == 0 ? (j - 1) + n*(k - 1) : c == 1 ? (i - 1) + n*(k - 1) : (i - 1) + n*(j - 1);
}

// predictor of the three components in one pass: gfs_face_advected_normal_velocity
// (src/advection.c:513-539) with the face values of u[c] along c computed in place
// (use_centered_velocity = TRUE)
template <bool VL, bool VS, bool MPI>
__global__ void __launch_bounds__(GN)
predict_un_tiled_kernel (Layout L, CPtr3 u, Ptr3 un, double dt, Visc3 visc, GhostFv G)
{
  __shared__ double fl[3][GN], fr[3][GN];
  __shared__ double hp[3][GX*GZ];          // r of the cell beyond the + face of the tile
  const TileIdx T;
  const int n = L.n;
  const int c = (int) L.idx (T.i, T.j, T.k);
  const int off[3] = { 1, (int) L.sy, (int) L.sz };
  CPtr3 none = { { nullptr, nullptr, nullptr } };
  {
    FacePair f = face_values_dir<3, 0, true, VL, VS> (L, u.p[0], u, none, c, dt, visc.d[0], visc.g[0]);
    fl[0][T.own ()] = f.l; fr[0][T.own ()] = f.r;
    f = face_values_dir<3, 1, true, VL, VS> (L, u.p[1], u, none, c, dt, visc.d[1], visc.g[1]);
    fl[1][T.own ()] = f.l; fr[1][T.own ()] = f.r;
    f = face_values_dir<3, 2, true, VL, VS> (L, u.p[2], u, none, c, dt, visc.d[2], visc.g[2]);
    fl[2][T.own ()] = f.l; fr[2][T.own ()] = f.r;
  }
  // halo: the cell beyond the + face of the tile in each direction.  Waves 0-3 take y, waves 4-7
  // z (one cell each, no divergence inside a wave), the first 16 lanes also x
  {
    const int h = threadIdx.x;
    if (h < GX*GZ) {
      int p = h % GX, q = h / GX;
      const int hi_ = blockIdx.x*GX + p + 1, hj = blockIdx.y*GY + GY + 1, hk = blockIdx.z*GZ + q + 1;
      if (MPI && G.r[2] && hj > n)        /* beyond an MPI side: the state the box after this one sent */
	hp[1][h] = G.r[2][face_pos (1, n, hi_, hj, hk)];
      else {
	int ci = image<3> (L, hi_, hj, hk);
	hp[1][h] = face_values_dir<3, 1, true, VL, VS> (L, u.p[1], u, none, ci, dt, visc.d[1], visc.g[1]).r;
      }
    }
    else if (h < GX*GZ + GX*GY) {
      int hh = h - GX*GZ, p = hh % GX, q = hh / GX;
      const int hi_ = blockIdx.x*GX + p + 1, hj = blockIdx.y*GY + q + 1, hk = blockIdx.z*GZ + GZ + 1;
      if (MPI && G.r[4] && hk > n)
	hp[2][hh] = G.r[4][face_pos (2, n, hi_, hj, hk)];
      else {
	int ci = image<3> (L, hi_, hj, hk);
	hp[2][hh] = face_values_dir<3, 2, true, VL, VS> (L, u.p[2], u, none, ci, dt, visc.d[2], visc.g[2]).r;
      }
    }
    if (h < GY*GZ) {
      int p = h % GY, q = h / GY;
      const int hi_ = blockIdx.x*GX + GX + 1, hj = blockIdx.y*GY + p + 1, hk = blockIdx.z*GZ + q + 1;
      if (MPI && G.r[0] && hi_ > n)
	hp[0][h] = G.r[0][face_pos (0, n, hi_, hj, hk)];
      else {
	int ci = image<3> (L, hi_, hj, hk);
	hp[0][h] = face_values_dir<3, 0, true, VL, VS> (L, u.p[0], u, none, ci, dt, visc.d[0], visc.g[0]).r;
      }
    }
  }
  __syncthreads ();
  const int t3[3] = { T.tx, T.ty, T.tz }, g3[3] = { GX, GY, GZ }, ijk[3] = { T.i, T.j, T.k };
  const int so[3] = { 1, GX, GX*GY };
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double * ud = u.p[d];
    // + face of the own cell
    double r;
    if (t3[d] + 1 < g3[d])
      r = fr[d][T.own () + so[d]];
    else
      r = hp[d][d == 0 ? T.ty + GY*T.tz : d == 1 ? T.tx + GX*T.tz : T.tx + GX*T.ty];
    double s_ = face_interp (ud[c], ud[c + off[d]]);
    double val = upwinded (s_, fl[d][T.own ()], r);
    un.p[d][c] = val;
    if (MPI && G.r[2*d + 1]) {
      // the face on the low side of the box: left state from the box before this one
      if (ijk[d] == 1) {
	const double gl = G.r[2*d + 1][face_pos (d, n, T.i, T.j, T.k)];
	const double s0 = face_interp (ud[c - off[d]], ud[c]);
	un.p[d][c - off[d]] = upwinded (s0, gl, fr[d][T.own ()]);
      }
    }
    // the face on the low side of the box is the periodic image of the one on the high side
    else if (ijk[d] == n)
      un.p[d][c - n*off[d]] = val;
  }
}

// variable_sources (src/timestep.c:872-921) of one advected variable: out = v + fluxes - g*dt,
// the fluxes gathered in the reference's scatter order (see flux_update_kernel)
template <bool VELOCITY, bool VL, bool VS>
__global__ void __launch_bounds__(GN)
advect_tiled_kernel (Layout L, const double * __restrict__ v, double * __restrict__ out, CPtr3 un,
		     const double * __restrict__ gm, const double * __restrict__ gc, double dt,
		     double visc, double gsrc)
{
  __shared__ double fl[3][GN], fr[3][GN];
  __shared__ double hm[3][GX*GZ], hp[3][GX*GZ];   // l of the cell before / r of the cell after the tile
  const TileIdx T;
  const int n = L.n;
  const double rn = (double) n;
  const int c = (int) L.idx (T.i, T.j, T.k);
  const int off[3] = { 1, (int) L.sy, (int) L.sz };
  CPtr3 none = { { nullptr, nullptr, nullptr } };
  {
    FacePair f = face_values_dir<3, 0, false, VL, VS> (L, v, none, un, c, dt, visc, gsrc);
    fl[0][T.own ()] = f.l; fr[0][T.own ()] = f.r;
    f = face_values_dir<3, 1, false, VL, VS> (L, v, none, un, c, dt, visc, gsrc);
    fl[1][T.own ()] = f.l; fr[1][T.own ()] = f.r;
    f = face_values_dir<3, 2, false, VL, VS> (L, v, none, un, c, dt, visc, gsrc);
    fl[2][T.own ()] = f.l; fr[2][T.own ()] = f.r;
  }
  // halo cells: l of the cell before the tile (minus side), r of the cell after it (plus side).
  // Waves 0-1: y minus, 2-3: y plus, 4-5: z minus, 6-7: z plus (one cell per lane, one direction
  // per wave); the first 32 lanes also take the x halos
  {
    const int h = threadIdx.x;
    const int grp = h / (GX*GZ), idx = h % (GX*GZ), p = idx % GX, q = idx / GX;
    if (grp < 2) {
      int ci = image<3> (L, blockIdx.x*GX + p + 1, blockIdx.y*GY + (grp ? GY + 1 : 0), blockIdx.z*GZ + q + 1);
      FacePair f = face_values_dir<3, 1, false, VL, VS> (L, v, none, un, ci, dt, visc, gsrc);
      if (grp) hp[1][idx] = f.r; else hm[1][idx] = f.l;
    }
    else {
      int ci = image<3> (L, blockIdx.x*GX + p + 1, blockIdx.y*GY + q + 1, blockIdx.z*GZ + (grp == 3 ? GZ + 1 : 0));
      FacePair f = face_values_dir<3, 2, false, VL, VS> (L, v, none, un, ci, dt, visc, gsrc);
      if (grp == 3) hp[2][idx] = f.r; else hm[2][idx] = f.l;
    }
    if (h < 2*GY*GZ) {
      const int plus = h >= GY*GZ, hh = h % (GY*GZ), py = hh % GY, qz = hh / GY;
      int ci = image<3> (L, blockIdx.x*GX + (plus ? GX + 1 : 0), blockIdx.y*GY + py + 1, blockIdx.z*GZ + qz + 1);
      FacePair f = face_values_dir<3, 0, false, VL, VS> (L, v, none, un, ci, dt, visc, gsrc);
      if (plus) hp[0][hh] = f.r; else hm[0][hh] = f.l;
    }
  }
  __syncthreads ();
  const int t3[3] = { T.tx, T.ty, T.tz }, g3[3] = { GX, GY, GZ };
  const int so[3] = { 1, GX, GX*GY };
  const int hi[3] = { T.ty + GY*T.tz, T.tx + GX*T.tz, T.tx + GX*T.ty };
  // flux through the + face (direction cf) of the own cell (minus = 0) or of the cell before it
  auto flux = [&] (int cf, int minus) -> double {
    int a = minus ? c - off[cf] : c;
    double l_, r_;
    if (!minus) {
      l_ = fl[cf][T.own ()];
      r_ = t3[cf] + 1 < g3[cf] ? fr[cf][T.own () + so[cf]] : hp[cf][hi[cf]];
    }
    else {
      l_ = t3[cf] > 0 ? fl[cf][T.own () - so[cf]] : hm[cf][hi[cf]];
      r_ = fr[cf][T.own ()];
    }
    double unf = un.p[cf][a];
    double upw = upwinded (unf, l_, r_);
    if (VELOCITY) {
      double f = 1.*unf*dt*rn;          /* /h, h = 1/n a power of two: exact scaling */
      f *= upw - face_interp (gm[a], gm[a + off[cf]])*dt/2.;
      return f;
    }
    return 1.*unf*dt*upw*rn;
  };
  const int i = T.i, j = T.j, k = T.k;
  double acc = 0.;
  if (i > 1)
    acc += flux (0, 1);
  acc -= flux (0, 0);
  acc -= flux (1, 0);
  acc -= flux (2, 0);
  {
    unsigned J = n - j, K = n - k;
    bool back_first = __ffs (~J) > __ffs (~K);
    if (back_first) {
      if (k > 1) acc += flux (2, 1);
      if (j > 1) acc += flux (1, 1);
    }
    else {
      if (j > 1) acc += flux (1, 1);
      if (k > 1) acc += flux (2, 1);
    }
  }
  if (i == 1)
    acc += flux (0, 1);
  if (j == 1)
    acc += flux (1, 1);
  if (k == 1)
    acc += flux (2, 1);
  double val = v[c];
  val += acc/1.;
  if (gc)
    val -= gc[c]*dt;
  if (VS && gsrc != 0.) {       /* gfs_domain_variable_centered_sources, src/source.c:62-108 */
    double sum = 0;
    sum += gsrc;
    val += dt*sum;
  }
  out[c] = val;
}

// ---------------------------------------------------------------------------------------------
// The three velocity components in ONE pass (gfs_centered_velocity_advection_diffusion without
// viscosity, src/timestep.c:976-1016: variable_sources of U, V, W with the same MAC velocities).
// advect_tiled_kernel is issue-bound, and most of what it issues does not depend on the advected
// variable: the MAC velocities of the cell and their sums (tangential velocities and CFL numbers of
// gfs_cell_advected_face_values), the upwind neighbours, the periodic images and addresses of the
// halo cells, the flux velocities un*dt/h of the faces, the gather order.  Here all of that is
// done once per cell for the three components.  Two more changes against the one-component kernel:
//   * only the `right' face values travel through LDS (the + face of a cell needs its own left
//     state and the right state of the cell after it);
//   * every face flux is computed once, by the cell on its - side, and handed to the cell on its
//     + side through LDS (same expression, same operands: the same bits the two cells used to
//     compute twice); the faces on the - sides of the tile are computed by the cells along them
//     from the left state of the cell before the tile.
// Same arithmetic and gather order as advect_tiled_kernel / flux_update_kernel: bit-identical.
// ---------------------------------------------------------------------------------------------
struct AdvShared {          // what gfs_cell_advected_face_values derives from the MAC velocities of a cell
  double vtan[3];           // (f[2c].un + f[2c+1].un)/2.
  double unorm[3];          // dt*(f[2c].un + f[2c+1].un)/(2.*size)
};

__device__ __forceinline__ AdvShared adv_shared (const Layout & L, const CPtr3 & un, int c, double dt)
{
  const int off[3] = { 1, (int) L.sy, (int) L.sz };
  const double rsize2 = (double) L.n/2.;
  AdvShared S;
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    const double a = un.p[cc][c], b = un.p[cc][c - off[cc]];
    S.vtan[cc] = (a + b)/2.;
    S.unorm[cc] = dt*(a + b)*rsize2;
  }
  return S;
}

// transverse term of direction cc (transverse_term, src/advection.c:27-47)
__device__ __forceinline__ double adv_transverse (const double * __restrict__ v, int c, int o,
						  double v0, double vtan, double dt, double rsize2)
{
  const int nb = vtan > 0. ? c - o : c + o;
  double g = v[nb] - 1.*v0;
  if (vtan > 0.) g = - g;
  return dt*vtan*g*rsize2;
}

// the same from the two neighbours along the direction, already loaded: the upwind neighbour is
// SELECTED, not loaded through an address that depends on the sign of vtan -- in the tiled kernels
// every load of a cell is then issued up front, independent of the MAC velocities (round 3: the
// counters of advect3_tiled_kernel showed 59 % of its wave cycles in s_waitcnt behind such chains
// of two dependent global loads, profiles/r03_pmc_advect3.json)
__device__ __forceinline__ double adv_transverse_v (double vminus, double vplus, double v0, double vtan,
						    double dt, double rsize2)
{
  double g = (vtan > 0. ? vminus : vplus) - 1.*v0;
  if (vtan > 0.) g = - g;
  return dt*vtan*g*rsize2;
}

// the seven values of the stencil of a cell
struct Stencil7 { double v0, m[3], p[3]; };

__device__ __forceinline__ Stencil7 load_stencil7 (const double * __restrict__ v, int c, const int off[3])
{
  Stencil7 S;
  S.v0 = v[c];
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    S.m[cc] = v[c - off[cc]];
    S.p[cc] = v[c + off[cc]];
  }
  return S;
}

template <int D, bool VL, bool SRC>
__device__ __forceinline__ FacePair adv_face_values_v (double v0, double v1, double v2, double unorm,
						       double ta, double tb, double dt, double gsrc);

// the two face values of direction D of a cell whose stencil is loaded (transverse terms of the other
// two directions in increasing order of direction)
template <int D, bool VL, bool SRC>
__device__ __forceinline__ FacePair adv_face_values_s (const Stencil7 & W, const AdvShared & S, double dt,
							double rsize2, double gsrc)
{
  constexpr int A = D == 0 ? 1 : 0, B = D == 2 ? 1 : 2;
  const double ta = adv_transverse_v (W.m[A], W.p[A], W.v0, S.vtan[A], dt, rsize2);
  const double tb = adv_transverse_v (W.m[B], W.p[B], W.v0, S.vtan[B], dt, rsize2);
  return adv_face_values_v<D, VL, SRC> (W.v0, W.m[D], W.p[D], S.unorm[D], ta, tb, dt, gsrc);
}

// the two face values of direction D given the transverse terms ta, tb of the other two directions
// (in increasing order of direction): face_values_dir with CEN = false, VS = false
template <int D, bool VL, bool SRC>
__device__ __forceinline__ FacePair adv_face_values (const double * __restrict__ v, int c, int o,
						     double v0, double unorm, double ta, double tb,
						     double dt, double gsrc)
{
  const double v1 = v[c - o], v2 = v[c + o];
  return adv_face_values_v<D, VL, SRC> (v0, v1, v2, unorm, ta, tb, dt, gsrc);
}

template <int D, bool VL, bool SRC>
__device__ __forceinline__ FacePair adv_face_values_v (double v0, double v1, double v2, double unorm,
						       double ta, double tb, double dt, double gsrc)
{
  const double g = VL ? van_leer_gradient (v0, v1, v2) : center_gradient (v0, v1, v2);
  const double vl = v0 + GMIN ((1. - unorm)/2., 0.5)*g;
  const double vr = v0 + GMAX ((- 1. - unorm)/2., -0.5)*g;
  double msrc = 0.;                      /* gfs_variable_mac_source: the intensity of a GfsSource, if any */
  if (SRC) msrc += gsrc;
  const double src = dt*msrc/2.;
  double dv = ta;
  dv += tb;
  FacePair f;
  f.l = vl + src - dv;
  f.r = vr + src - dv;
  return f;
}

// ---------------------------------------------------------------------------------------------
// The same kernels on a box with GfsBoundaryMpi sides.  The face value of a ghost cell beyond an MPI
// side is what the neighbour box computed for its cell along that side (gfs_domain_face_bc over
// gfs_boundary_send / receive): of everything the six face-value arrays hold, a box needs from its
// neighbours one number per ghost cell and advected variable -- the left state of the layer n of the
// box before it, the right state of the layer 1 of the box after it, in the direction of the side.
// boundary_face_values_kernel computes exactly those for the own layers (the expressions of the
// tiled kernels, hence the bits the neighbour would have found in the exchanged arrays), they
// travel in ONE message per side (3 n^2 doubles for the three velocity components instead of the
// six exchanges of whole-face arrays per component), and the tiled kernels take the states of the
// cells before / after the box from the received buffers instead of recomputing periodic images.
//   send[2c]   [q n^2 + f] = left state (direction c) of variable q in the cell of layer n at face position f
//   send[2c+1] [q n^2 + f] = right state ........................................ layer 1
//   recv[2c]: the right states of the layer 1 of the box across side 2c; recv[2c+1]: the left states
//   of the layer n of the box across side 2c + 1
// f = (x_a - 1) + n (x_b - 1) with a < b the two directions other than c.
// ---------------------------------------------------------------------------------------------
// CEN: the predictor (one variable per direction: u[c] along c, centred velocities); otherwise the
// three velocity components with the MAC velocities (advect3_tiled_kernel)
template <bool CEN, bool VL, bool VS>
__global__ void __launch_bounds__(256)
boundary_face_values_kernel (Layout L, CPtr3 v, CPtr3 un, double dt, Visc3 visc, GhostFv G)
{
  const int n = L.n;
  const int f = blockIdx.x*blockDim.x + threadIdx.x;
  const int d = blockIdx.y;
  if (f >= n*n || !G.s[d]) return;
  const int c = d/2;
  const int t1 = f % n + 1, t2 = f / n + 1;
  int ijk[3];
  ijk[c] = (d & 1) ? 1 : n;
  ijk[c == 0 ? 1 : 0] = t1;
  ijk[c == 2 ? 1 : 2] = t2;
  const int ci = (int) L.idx (ijk[0], ijk[1], ijk[2]);
  const int off[3] = { 1, (int) L.sy, (int) L.sz };
  const size_t nf = (size_t) n*n;
  if (CEN) {
    CPtr3 none = { { nullptr, nullptr, nullptr } };
    FacePair p;
    if (c == 0)      p = face_values_dir<3, 0, true, VL, VS> (L, v.p[0], v, none, ci, dt, visc.d[0], visc.g[0]);
    else if (c == 1) p = face_values_dir<3, 1, true, VL, VS> (L, v.p[1], v, none, ci, dt, visc.d[1], visc.g[1]);
    else             p = face_values_dir<3, 2, true, VL, VS> (L, v.p[2], v, none, ci, dt, visc.d[2], visc.g[2]);
    G.s[d][f] = (d & 1) ? p.r : p.l;
  }
  else {
    const double rsize2 = (double) n/2.;
    const AdvShared S = adv_shared (L, un, ci, dt);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const double * vq = v.p[q];
      const double w0 = vq[ci];
      FacePair p;
      if (c == 0) {
	const double ta = adv_transverse (vq, ci, off[1], w0, S.vtan[1], dt, rsize2);
	const double tb = adv_transverse (vq, ci, off[2], w0, S.vtan[2], dt, rsize2);
	p = adv_face_values<0, VL, VS> (vq, ci, off[0], w0, S.unorm[0], ta, tb, dt, visc.g[q]);
      }
      else if (c == 1) {
	const double ta = adv_transverse (vq, ci, off[0], w0, S.vtan[0], dt, rsize2);
	const double tb = adv_transverse (vq, ci, off[2], w0, S.vtan[2], dt, rsize2);
	p = adv_face_values<1, VL, VS> (vq, ci, off[1], w0, S.unorm[1], ta, tb, dt, visc.g[q]);
      }
      else {
	const double ta = adv_transverse (vq, ci, off[0], w0, S.vtan[0], dt, rsize2);
	const double tb = adv_transverse (vq, ci, off[1], w0, S.vtan[1], dt, rsize2);
	p = adv_face_values<2, VL, VS> (vq, ci, off[2], w0, S.unorm[2], ta, tb, dt, visc.g[q]);
      }
      G.s[d][q*nf + f] = (d & 1) ? p.r : p.l;
    }
  }
}

// WPE: waves per SIMD the register allocation is held to (1 = no constraint)
// CORR: gfs_correct_centered_velocities (src/timestep.c:498-530) with the same gradient and - dt, the
// reference's next operation on U, V, W in simulation_run (src/simulation.c:519-525), applied to the
// value before it is stored -- u = (v + fluxes - g dt) - g (- dt), the two roundings of the two
// passes -- and, with uc, the first level of gfs_cell_coarse_init of the corrected velocities
// (correct_centered_coarse_kernel's expressions; a tile holds whole groups of eight children).  The
// BC application between the two operations of the reference writes ghost cells only, which the one
// after the correction rewrites.
// LDS: 49.9 KB per tile (the x halos in arrays of their own size): three tiles per CU instead of two.
struct AdvCorr { double dt; double * uc[3]; Layout Lc; };
#ifndef ADV3_STENCIL
#define ADV3_STENCIL 0     /* 1: the own cell selects its upwind neighbours from the loaded stencil (measured slower: 1.48 against 1.29 ms) */
#endif
#ifndef ADV3_WPE
#define ADV3_WPE 4     /* two tiles of 512 threads per CU: 4 waves per SIMD, 128 VGPRs (80 for three tiles spills) */
#endif

template <bool VL, bool MPI, bool SRC, int WPE, bool CORR>
__global__ void __launch_bounds__(GN, WPE)
advect3_tiled_kernel (Layout L, CPtr3 v, Ptr3 out, CPtr3 un, CPtr3 gm, CPtr3 gc, double dt, GhostFv G,
		      Visc3 src3, AdvCorr K)
{
  // R: right face values, then (after they have been read) the fluxes of the + faces, then (CORR)
  // the new values for the coarse cells
  __shared__ double R[3][3][GN];
  // l of the cell before / r of the cell after the tile: y and z halos, x halos
  __shared__ double hm[3][2][GX*GZ], hp[3][2][GX*GZ], hmx[3][GY*GZ], hpx[3][GY*GZ];
  const TileIdx T;
  const int n = L.n;
  const double rn = (double) n, rsize2 = (double) n/2.;
  const int c = (int) L.idx (T.i, T.j, T.k);
  const int off[3] = { 1, (int) L.sy, (int) L.sz };
  const int own = T.own ();
  double v0[3], fl[3][3], fr[3][3];
  {
    const AdvShared S = adv_shared (L, un, c, dt);
#pragma unroll
    for (int q = 0; q < 3; q++) {
#if ADV3_STENCIL
      // the six neighbours are needed for the gradients anyway: the upwind ones are selected from them
      const Stencil7 W = load_stencil7 (v.p[q], c, off);
      v0[q] = W.v0;
      FacePair f = adv_face_values_s<0, VL, SRC> (W, S, dt, rsize2, src3.g[q]);
      fl[q][0] = f.l; fr[q][0] = f.r;
      f = adv_face_values_s<1, VL, SRC> (W, S, dt, rsize2, src3.g[q]);
      fl[q][1] = f.l; fr[q][1] = f.r;
      f = adv_face_values_s<2, VL, SRC> (W, S, dt, rsize2, src3.g[q]);
      fl[q][2] = f.l; fr[q][2] = f.r;
#else
      const double * vq = v.p[q];
      v0[q] = vq[c];
      double t[3];
#pragma unroll
      for (int cc = 0; cc < 3; cc++)
	t[cc] = adv_transverse (vq, c, off[cc], v0[q], S.vtan[cc], dt, rsize2);
      FacePair f = adv_face_values<0, VL, SRC> (vq, c, off[0], v0[q], S.unorm[0], t[1], t[2], dt, src3.g[q]);
      fl[q][0] = f.l; fr[q][0] = f.r;
      f = adv_face_values<1, VL, SRC> (vq, c, off[1], v0[q], S.unorm[1], t[0], t[2], dt, src3.g[q]);
      fl[q][1] = f.l; fr[q][1] = f.r;
      f = adv_face_values<2, VL, SRC> (vq, c, off[2], v0[q], S.unorm[2], t[0], t[1], dt, src3.g[q]);
      fl[q][2] = f.l; fr[q][2] = f.r;
#endif
#pragma unroll
      for (int d = 0; d < 3; d++)
	R[q][d][own] = fr[q][d];
    }
  }
  // halo cells, one direction each: waves 0-1 y minus, 2-3 y plus, 4-5 z minus, 6-7 z plus (one cell
  // per lane); the first 32 lanes also take the x halos
  {
    const int h = threadIdx.x;
    const int grp = h / (GX*GZ), idx = h % (GX*GZ), p = idx % GX, qq = idx / GX;
    const size_t nf = (size_t) n*n;
    if (grp < 2) {
      const int hi_ = blockIdx.x*GX + p + 1, hj = blockIdx.y*GY + (grp ? GY + 1 : 0), hk = blockIdx.z*GZ + qq + 1;
      const double * gs = grp ? G.r[2] : G.r[3];     /* constant indices: G stays in scalar registers */
      if (MPI && gs && (hj < 1 || hj > n)) {
	// beyond an MPI side: the states the neighbour box sent
	const double * g = gs + face_pos (1, n, hi_, hj, hk);
#pragma unroll
	for (int q = 0; q < 3; q++)
	  if (grp) hp[q][0][idx] = g[q*nf]; else hm[q][0][idx] = g[q*nf];
      }
      else {
	const int ci = image<3> (L, hi_, hj, hk);
	const AdvShared S = adv_shared (L, un, ci, dt);
#pragma unroll
	for (int q = 0; q < 3; q++) {
	  const double * vq = v.p[q];
	  const double w0 = vq[ci];
	  const double ta = adv_transverse (vq, ci, off[0], w0, S.vtan[0], dt, rsize2);
	  const double tb = adv_transverse (vq, ci, off[2], w0, S.vtan[2], dt, rsize2);
	  const FacePair f = adv_face_values<1, VL, SRC> (vq, ci, off[1], w0, S.unorm[1], ta, tb, dt, src3.g[q]);
	  if (grp) hp[q][0][idx] = f.r; else hm[q][0][idx] = f.l;
	}
      }
    }
    else {
      const int hi_ = blockIdx.x*GX + p + 1, hj = blockIdx.y*GY + qq + 1, hk = blockIdx.z*GZ + (grp == 3 ? GZ + 1 : 0);
      const double * gs = grp == 3 ? G.r[4] : G.r[5];
      if (MPI && gs && (hk < 1 || hk > n)) {
	const double * g = gs + face_pos (2, n, hi_, hj, hk);
#pragma unroll
	for (int q = 0; q < 3; q++)
	  if (grp == 3) hp[q][1][idx] = g[q*nf]; else hm[q][1][idx] = g[q*nf];
      }
      else {
	const int ci = image<3> (L, hi_, hj, hk);
	const AdvShared S = adv_shared (L, un, ci, dt);
#pragma unroll
	for (int q = 0; q < 3; q++) {
	  const double * vq = v.p[q];
	  const double w0 = vq[ci];
	  const double ta = adv_transverse (vq, ci, off[0], w0, S.vtan[0], dt, rsize2);
	  const double tb = adv_transverse (vq, ci, off[1], w0, S.vtan[1], dt, rsize2);
	  const FacePair f = adv_face_values<2, VL, SRC> (vq, ci, off[2], w0, S.unorm[2], ta, tb, dt, src3.g[q]);
	  if (grp == 3) hp[q][1][idx] = f.r; else hm[q][1][idx] = f.l;
	}
      }
    }
    if (h < 2*GY*GZ) {
      const int plus = h >= GY*GZ, hh = h % (GY*GZ), py = hh % GY, qz = hh / GY;
      const int hi_ = blockIdx.x*GX + (plus ? GX + 1 : 0), hj = blockIdx.y*GY + py + 1, hk = blockIdx.z*GZ + qz + 1;
      const double * gs = plus ? G.r[0] : G.r[1];
      if (MPI && gs && (hi_ < 1 || hi_ > n)) {
	const double * g = gs + face_pos (0, n, hi_, hj, hk);
#pragma unroll
	for (int q = 0; q < 3; q++)
	  if (plus) hpx[q][hh] = g[q*nf]; else hmx[q][hh] = g[q*nf];
      }
      else {
	const int ci = image<3> (L, hi_, hj, hk);
	const AdvShared S = adv_shared (L, un, ci, dt);
#pragma unroll
	for (int q = 0; q < 3; q++) {
	  const double * vq = v.p[q];
	  const double w0 = vq[ci];
	  const double ta = adv_transverse (vq, ci, off[1], w0, S.vtan[1], dt, rsize2);
	  const double tb = adv_transverse (vq, ci, off[2], w0, S.vtan[2], dt, rsize2);
	  const FacePair f = adv_face_values<0, VL, SRC> (vq, ci, off[0], w0, S.unorm[0], ta, tb, dt, src3.g[q]);
	  if (plus) hpx[q][hh] = f.r; else hmx[q][hh] = f.l;
	}
      }
    }
  }
  // what the flux phase reads from memory, issued before the barrier
  double unf[3], gm0[3], gmp[3][3], gcv[3];
#pragma unroll
  for (int d = 0; d < 3; d++)
    unf[d] = un.p[d][c];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    gm0[q] = gm.p[q][c];
#pragma unroll
    for (int d = 0; d < 3; d++)
      gmp[q][d] = gm.p[q][c + off[d]];
    gcv[q] = gc.p[q] ? gc.p[q][c] : 0.;
  }
  __syncthreads ();
  const int t3[3] = { T.tx, T.ty, T.tz }, g3[3] = { GX, GY, GZ };
  const int so[3] = { 1, GX, GX*GY };
  const int hi[3] = { T.ty + GY*T.tz, T.tx + GX*T.tz, T.tx + GX*T.ty };
  // fluxes through the + faces of the own cell (gfs_face_velocity_advection_flux,
  // src/advection.c:398-435): flux = 1.*un*dt/h; flux *= upwinded value - interpolated g*dt/2.
  double Fp[3][3], rnb[3][3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    rnb[q][0] = t3[0] + 1 < g3[0] ? R[q][0][own + so[0]] : hpx[q][hi[0]];
    rnb[q][1] = t3[1] + 1 < g3[1] ? R[q][1][own + so[1]] : hp[q][0][hi[1]];
    rnb[q][2] = t3[2] + 1 < g3[2] ? R[q][2][own + so[2]] : hp[q][1][hi[2]];
  }
  __syncthreads ();          // every right value has been read: R now takes the fluxes
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double unf = un.p[d][c];
    const double fu = 1.*unf*dt*rn;          /* /h, h = 1/n a power of two: exact scaling */
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const double upw = upwinded (unf, fl[q][d], rnb[q][d]);
      double f = fu;
      f *= upw - face_interp (gm.p[q][c], gm.p[q][c + off[d]])*dt/2.;
      Fp[q][d] = f;
      R[q][d][own] = f;
    }
  }
  __syncthreads ();
  // fluxes through the - faces: from the cell before, or computed here along the - sides of the tile
  double Fm[3][3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    if (t3[d] > 0) {
#pragma unroll
      for (int q = 0; q < 3; q++)
	Fm[q][d] = R[q][d][own - so[d]];
    }
    else {
      const int a = c - off[d];
      const double ua = un.p[d][a];
      const double fu = 1.*ua*dt*rn;
#pragma unroll
      for (int q = 0; q < 3; q++) {
	const double lm = d == 0 ? hmx[q][hi[0]] : hm[q][d - 1][hi[d]];
	const double upw = upwinded (ua, lm, fr[q][d]);
	double f = fu;
	f *= upw - face_interp (gm.p[q][a], gm.p[q][c])*dt/2.;
	Fm[q][d] = f;
      }
    }
  }
  // the gather in the reference's scatter order (flux_update_kernel)
  const int i = T.i, j = T.j, k = T.k;
  const unsigned J = n - j, K_ = n - k;
  const bool back_first = __ffs (~J) > __ffs (~K_);
  double res[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    double acc = 0.;
    if (i > 1)
      acc += Fm[q][0];
    acc -= Fp[q][0];
    acc -= Fp[q][1];
    acc -= Fp[q][2];
    if (back_first) {
      if (k > 1) acc += Fm[q][2];
      if (j > 1) acc += Fm[q][1];
    }
    else {
      if (j > 1) acc += Fm[q][1];
      if (k > 1) acc += Fm[q][2];
    }
    if (i == 1)
      acc += Fm[q][0];
    if (j == 1)
      acc += Fm[q][1];
    if (k == 1)
      acc += Fm[q][2];
    double val = v0[q];
    val += acc/1.;
    const double gcv = gc.p[q] ? gc.p[q][c] : 0.;
    if (gc.p[q])
      val -= gcv*dt;
    if (SRC && src3.g[q] != 0.) { /* gfs_domain_variable_centered_sources, src/source.c:62-108 */
      double sum = 0;
      sum += src3.g[q];
      val += dt*sum;
    }
    if (CORR)          /* correct (src/timestep.c:486-496): u[c] -= g[c]*dt with dt = K.dt */
      val = val - gcv*K.dt;
    out.p[q][c] = val;
    res[q] = val;
  }
  if (CORR && K.uc[0]) {
    // gfs_get_from_below_intensive of the corrected velocities (coarse_init_kernel): the eight
    // children of a coarse cell are cells of this tile; children in child-id order
    __syncthreads ();        // the fluxes in R have been read
#pragma unroll
    for (int q = 0; q < 3; q++)
      R[q][0][own] = res[q];
    __syncthreads ();
    if (threadIdx.x < GN/8) {
      const int cx = threadIdx.x % (GX/2), cy = (threadIdx.x / (GX/2)) % (GY/2), cz = threadIdx.x / ((GX/2)*(GY/2));
      const int pi = blockIdx.x*(GX/2) + cx + 1, pj = blockIdx.y*(GY/2) + cy + 1, pk = blockIdx.z*(GZ/2) + cz + 1;
#pragma unroll
      for (int q = 0; q < 3; q++) {
	double val = 0., sa = 0.;
#pragma unroll
	for (int id = 0; id < 8; id++) {
	  // ci = 2 pi - 1 + (id & 1), cj = 2 pj - ((id >> 1) & 1), ck = 2 pk - ((id >> 2) & 1): local indices
	  const int lx = 2*cx + (id & 1), ly = 2*cy + 1 - ((id >> 1) & 1), lz = 2*cz + 1 - ((id >> 2) & 1);
	  val += R[q][0][lx + GX*(ly + GY*lz)]*1.;
	  sa += 1.;
	}
	K.uc[q][K.Lc.idx (pi, pj, pk)] = val/sa;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The three velocity components advected by a sweep along z (periodic boxes): a workgroup owns a
// column of SWX x SWY cells and marches over SWZ planes, keeping what the planes share in registers
// and the neighbours inside a plane in LDS.
//
// Why (round 3, profiles/r03_pmc_advect3.json): the 32 x 4 x 4 tiles of advect3_tiled_kernel issue 77
// global loads per cell (every stencil value of every component through the vector cache, whose 32 KB
// hold a fraction of the 13 arrays x 34 x 6 x 6 cells of two resident tiles), evaluate 1.06 halo cells
// per cell and spend 59 % of their wave cycles waiting for memory with the vector ALUs 36 % busy.
// Here a cell loads its own values once per array (v, un, gm: the plane ahead; gc), its x / y
// neighbours come from a plane buffer in LDS, its z neighbours are the registers of the planes before
// and after, the face values of a plane are computed once and the flux through a z face is handed to
// the next plane in a register; the only cells evaluated twice are the ring around the column:
// 2 (SWX + SWY) of SWX SWY per plane, by 96 dedicated lanes of the first two waves.
//
// Pipeline, plane p = kb .. kb + SWZ + 1 (the planes kb and kb + SWZ + 1 only feed their neighbours):
//   A (p)      face values fl, fr [component][direction] of every cell of plane p
//   B (p - 1)  fluxes through the + faces of plane p - 1: x, y from the right states of the neighbours
//              (LDS), z from fr_z (p) of the same thread
//   C (p - 1)  the gather in the reference's scatter order, the update, the store
// Same expressions as advect3_tiled_kernel (adv_face_values_v, adv_transverse_v, upwinded,
// face_interp): bit-identical results (tests/test_gpu_timestep.py).
// ---------------------------------------------------------------------------------------------
#define SWX 32
#ifndef SWY
#define SWY 16
#endif
#define SWN (SWX*SWY)
#ifndef SWZ
#define SWZ 64
#endif

struct SweepFv { double l[3], r[3]; };

// the six face values of a cell from its stencil and the MAC velocities around it
template <bool VL, bool SRC>
__device__ __forceinline__ SweepFv sweep_face_values (const Stencil7 & W, const AdvShared & S, double dt,
						      double rsize2, double gsrc)
{
  SweepFv F;
  FacePair f = adv_face_values_s<0, VL, SRC> (W, S, dt, rsize2, gsrc);
  F.l[0] = f.l; F.r[0] = f.r;
  f = adv_face_values_s<1, VL, SRC> (W, S, dt, rsize2, gsrc);
  F.l[1] = f.l; F.r[1] = f.r;
  f = adv_face_values_s<2, VL, SRC> (W, S, dt, rsize2, gsrc);
  F.l[2] = f.l; F.r[2] = f.r;
  return F;
}

__device__ __forceinline__ AdvShared adv_shared_v (const double a[3], const double b[3], double dt, double rsize2)
{
  AdvShared S;
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    S.vtan[cc] = (a[cc] + b[cc])/2.;
    S.unorm[cc] = dt*(a[cc] + b[cc])*rsize2;
  }
  return S;
}

// LDS of a workgroup
struct SweepLds {
  double V[3][SWY + 2][SWX + 2];          // plane p of the three components, with the ring
  double UNx[SWY][SWX + 1], UNy[SWY + 1][SWX];   // un_x with the column before, un_y with the row before
  double FRx[3][SWY][SWX + 1], FRy[3][SWY + 1][SWX];   // right states of plane p - 1 (+ ring after)
  double FPx[3][SWY][SWX + 1], FPy[3][SWY + 1][SWX];   // + face fluxes of plane p - 1 (+ ring before)
};

#define SW_RING 128      /* two more waves: the ring cells (96 lanes) */

// The ring waves: lane rid = 0..31 the row before the column (y -), 32..63 the row after (y +), 64..79
// the column before (x -), 80..95 the column after (x +); one cell each, its face values in ONE
// direction.  The - lanes also compute the flux through the face between their cell and the column.
// Same loop and the same two barriers per plane as the compute waves.
template <bool VL, bool SRC>
__device__ __forceinline__ void sweep_ring_path (SweepLds & S_, const Layout & L, const CPtr3 & v, const CPtr3 & un,
						 const CPtr3 & gm, double dt, const Visc3 & src3, int rid)
{
  const int n = L.n;
  const double rn = (double) n, rsize2 = (double) n/2.;
  const int sy = (int) L.sy, sz = (int) L.sz;
  const int kb = blockIdx.z*SWZ;
  auto plane = [n, sz] (int k) { return (k < 1 ? k + n : k > n ? k - n : k)*sz; };
  int role = -1, hd = 0, hcol = 0, hslot = 0;
  if (rid < 2*SWX) {
    role = rid < SWX ? 0 : 1; hd = 1; hslot = rid % SWX;
    int hj = blockIdx.y*SWY + (role ? SWY + 1 : 0);
    hj = hj < 1 ? hj + n : hj > n ? hj - n : hj;
    hcol = (int) L.idx (blockIdx.x*SWX + hslot + 1, hj, 0);
  }
  else if (rid < 2*SWX + 2*SWY) {
    role = rid < 2*SWX + SWY ? 2 : 3; hd = 0; hslot = (rid - 2*SWX) % SWY;
    int hi_ = blockIdx.x*SWX + (role == 3 ? SWX + 1 : 0);
    hi_ = hi_ < 1 ? hi_ + n : hi_ > n ? hi_ - n : hi_;
    hcol = (int) L.idx (hi_, blockIdx.y*SWY + hslot + 1, 0);
  }
  const bool ring = role >= 0, ring_minus = role == 0 || role == 2;
  const int hoff = hd == 0 ? 1 : sy;                 // towards the column
  const int ry = role == 0 ? 0 : role == 1 ? SWY + 1 : hslot + 1;
  const int rx = role == 2 ? 0 : role == 3 ? SWX + 1 : hslot + 1;
  double hvm[3], hv0[3] = { 0., 0., 0. }, hvp[3] = { 0., 0., 0. }, hvn[3] = { 0., 0., 0. };
  double hunb = 0., hunzb = 0., hlo[3] = { 0., 0., 0. }, hgmo[3] = { 0., 0., 0. };
  if (ring) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      hv0[q] = v.p[q][hcol + plane (kb - 1)];
      hvp[q] = v.p[q][hcol + plane (kb)];
      hvn[q] = v.p[q][hcol + plane (kb + 1)];
    }
    hunzb = un.p[2][hcol + plane (kb - 1)];
  }
  for (int p = kb; p <= kb + SWZ + 1; p++) {
    const int zp = plane (p);
    Stencil7 HW[3];
    double hua[3] = { 0., 0., 0. }, hub[3] = { 0., 0., 0. };
#pragma unroll
    for (int q = 0; q < 3; q++) {
      hvm[q] = hv0[q]; hv0[q] = hvp[q]; hvp[q] = hvn[q];
      HW[q].v0 = hv0[q];
      HW[q].m[0] = HW[q].p[0] = HW[q].m[1] = HW[q].p[1] = 0.;
      HW[q].m[2] = hvm[q]; HW[q].p[2] = hvp[q];
    }
    if (ring) {
      const int hc = hcol + zp;
#pragma unroll
      for (int q = 0; q < 3; q++) {
	hvn[q] = v.p[q][hcol + plane (p + 2)];
	HW[q].m[0] = v.p[q][hc - 1];  HW[q].p[0] = v.p[q][hc + 1];
	HW[q].m[1] = v.p[q][hc - sy]; HW[q].p[1] = v.p[q][hc + sy];
      }
      hua[0] = un.p[0][hc]; hub[0] = un.p[0][hc - 1];
      hua[1] = un.p[1][hc]; hub[1] = un.p[1][hc - sy];
      hua[2] = un.p[2][hc]; hub[2] = hunzb;
#pragma unroll
      for (int q = 0; q < 3; q++)
	S_.V[q][ry][rx] = hv0[q];
      if (role == 2) S_.UNx[hslot][0] = hua[0];
      if (role == 0) S_.UNy[0][hslot] = hua[1];
    }
    __syncthreads ();                                // (1)
    double hl[3] = { 0., 0., 0. }, hr[3] = { 0., 0., 0. };
    if (ring) {
      const AdvShared S = adv_shared_v (hua, hub, dt, rsize2);
#pragma unroll
      for (int q = 0; q < 3; q++) {
	const FacePair f = hd == 0 ? adv_face_values_s<0, VL, SRC> (HW[q], S, dt, rsize2, src3.g[q]) :
	  adv_face_values_s<1, VL, SRC> (HW[q], S, dt, rsize2, src3.g[q]);
	hl[q] = f.l; hr[q] = f.r;
      }
      // the face between the cell and the column, plane p - 1: by the - lanes, which hold the left state
      // of their cell and read the right state of the cell of the column
      if (p > kb && ring_minus) {
	const int ho = hcol + plane (p - 1);
	const double ua = hunb;
	const double fu = 1.*ua*dt*rn;
#pragma unroll
	for (int q = 0; q < 3; q++) {
	  const double rs = role == 2 ? S_.FRx[q][hslot][0] : S_.FRy[q][0][hslot];
	  double f = fu;
	  f *= upwinded (ua, hlo[q], rs) - face_interp (hgmo[q], gm.p[q][ho + hoff])*dt/2.;
	  if (role == 2) S_.FPx[q][hslot][0] = f; else S_.FPy[q][0][hslot] = f;
	}
      }
    }
    __syncthreads ();                                // (2)
    if (ring) {
#pragma unroll
      for (int q = 0; q < 3; q++) {
	if (role == 3) S_.FRx[q][hslot][SWX] = hr[q];
	if (role == 1) S_.FRy[q][SWY][hslot] = hr[q];
	hlo[q] = hl[q];
	hgmo[q] = gm.p[q][hcol + zp];
      }
      hunb = hua[hd];
      hunzb = hua[2];
    }
  }
}

template <bool VL, bool SRC, bool CORR>
__global__ void __launch_bounds__(SWN + SW_RING)
advect3_sweep_kernel (Layout L, CPtr3 v, Ptr3 out, CPtr3 un, CPtr3 gm, CPtr3 gc, double dt, Visc3 src3,
		      AdvCorr K)
{
  __shared__ SweepLds S_;
  const int tid = threadIdx.x;
  if (tid >= SWN) {
    sweep_ring_path<VL, SRC> (S_, L, v, un, gm, dt, src3, tid - SWN);
    return;
  }
  const int tx = tid % SWX, ty = tid / SWX;
  const int n = L.n;
  const double rn = (double) n, rsize2 = (double) n/2.;
  const int sy = (int) L.sy, sz = (int) L.sz;
  const int i = blockIdx.x*SWX + tx + 1, j = blockIdx.y*SWY + ty + 1;
  const int kb = blockIdx.z*SWZ;                     // output planes kb + 1 .. kb + SWZ
  const int col = (int) L.idx (i, j, 0);
  auto plane = [n, sz] (int k) { return (k < 1 ? k + n : k > n ? k - n : k)*sz; };   // periodic image

  // ---- registers of the pipeline
  double vm[3], v0[3], vp[3], vn[3];                 // planes p - 1, p, p + 1 and the load for p + 2
  double una[3], unb[3];                             // un (p) of the own cell; un (p - 1) (B of plane p - 1)
  double flo[3][3];                                  // left states of plane p - 1
  double fpz[3] = { 0., 0., 0. };                    // fluxes through the z face below plane p - 1
  double gmo[3];                                     // gm (p - 1) of the own cell
  double low[3] = { 0., 0., 0. };                    // corrected values of the odd plane (coarse cells)
#pragma unroll
  for (int q = 0; q < 3; q++) {
    v0[q] = v.p[q][col + plane (kb - 1)];
    vp[q] = v.p[q][col + plane (kb)];
    vn[q] = v.p[q][col + plane (kb + 1)];
    gmo[q] = gm.p[q][col + plane (kb)];       /* the first B is that of plane kb */
#pragma unroll
    for (int d = 0; d < 3; d++) flo[q][d] = 0.;
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
    una[d] = un.p[d][col + plane (kb - 1)];
    unb[d] = 0.;
  }

  for (int p = kb; p <= kb + SWZ + 1; p++) {
    const int zp = plane (p);
    // ---- shift the column registers; the loads of this iteration
#pragma unroll
    for (int q = 0; q < 3; q++) {
      vm[q] = v0[q]; v0[q] = vp[q]; vp[q] = vn[q];
      vn[q] = v.p[q][col + plane (p + 2)];
    }
    const double unzb = una[2];                      // un_z (p - 1)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      unb[d] = una[d];
      una[d] = un.p[d][col + zp];
    }
    // ---- the plane into LDS
#pragma unroll
    for (int q = 0; q < 3; q++)
      S_.V[q][ty + 1][tx + 1] = v0[q];
    S_.UNx[ty][tx + 1] = una[0];
    S_.UNy[ty + 1][tx] = una[1];
    __syncthreads ();                                // (1) plane p is there; nothing of FP is read any more
    // ---- A (p)
    SweepFv F[3];
    {
      const double b[3] = { S_.UNx[ty][tx], S_.UNy[ty][tx], unzb };
      const AdvShared S = adv_shared_v (una, b, dt, rsize2);
#pragma unroll
      for (int q = 0; q < 3; q++) {
	Stencil7 W;
	W.v0 = v0[q];
	W.m[0] = S_.V[q][ty + 1][tx];     W.p[0] = S_.V[q][ty + 1][tx + 2];
	W.m[1] = S_.V[q][ty][tx + 1];     W.p[1] = S_.V[q][ty + 2][tx + 1];
	W.m[2] = vm[q];                   W.p[2] = vp[q];
	F[q] = sweep_face_values<VL, SRC> (W, S, dt, rsize2, src3.g[q]);
      }
    }
    double Fp[3][3];
    if (p > kb) {
      // ---- B (o): fluxes through the + faces of plane o = p - 1
      const int co = col + plane (p - 1);
      const double fux = 1.*unb[0]*dt*rn, fuy = 1.*unb[1]*dt*rn, fuz = 1.*unb[2]*dt*rn;
#pragma unroll
      for (int q = 0; q < 3; q++) {
	const double g0 = gmo[q];
	const double gx = gm.p[q][co + 1], gy = gm.p[q][co + sy], gz = gm.p[q][col + zp];
	double f = fux;
	f *= upwinded (unb[0], flo[q][0], S_.FRx[q][ty][tx + 1]) - face_interp (g0, gx)*dt/2.;
	Fp[q][0] = f;
	f = fuy;
	f *= upwinded (unb[1], flo[q][1], S_.FRy[q][ty + 1][tx]) - face_interp (g0, gy)*dt/2.;
	Fp[q][1] = f;
	f = fuz;
	f *= upwinded (unb[2], flo[q][2], F[q].r[2]) - face_interp (g0, gz)*dt/2.;
	Fp[q][2] = f;
	gmo[q] = gz;
	S_.FPx[q][ty][tx + 1] = Fp[q][0];
	S_.FPy[q][ty + 1][tx] = Fp[q][1];
      }
    }
    __syncthreads ();                                // (2) the fluxes of plane p - 1; plane p has been read
    if (p > kb) {
      const int o = p - 1, co = col + plane (o);
      if (p > kb + 1) {
	// ---- C (o): the gather in the reference's scatter order (flux_update_kernel), the update
	const int k = o;
	const unsigned J = n - j, K_ = n - k;
	const bool back_first = __ffs (~J) > __ffs (~K_);
	double res[3];
#pragma unroll
	for (int q = 0; q < 3; q++) {
	  const double Fmx = S_.FPx[q][ty][tx], Fmy = S_.FPy[q][ty][tx], Fmz = fpz[q];
	  double acc = 0.;
	  if (i > 1)
	    acc += Fmx;
	  acc -= Fp[q][0];
	  acc -= Fp[q][1];
	  acc -= Fp[q][2];
	  if (back_first) {
	    if (k > 1) acc += Fmz;
	    if (j > 1) acc += Fmy;
	  }
	  else {
	    if (j > 1) acc += Fmy;
	    if (k > 1) acc += Fmz;
	  }
	  if (i == 1)
	    acc += Fmx;
	  if (j == 1)
	    acc += Fmy;
	  if (k == 1)
	    acc += Fmz;
	  double val = vm[q];
	  val += acc/1.;
	  const double gcv = gc.p[q] ? gc.p[q][co] : 0.;
	  if (gc.p[q])
	    val -= gcv*dt;
	  if (SRC && src3.g[q] != 0.) { /* gfs_domain_variable_centered_sources, src/source.c:62-108 */
	    double sum = 0;
	    sum += src3.g[q];
	    val += dt*sum;
	  }
	  if (CORR)          /* correct (src/timestep.c:486-496): u[c] -= g[c]*dt with dt = K.dt */
	    val = val - gcv*K.dt;
	  out.p[q][co] = val;
	  res[q] = val;
	}
	if (CORR && K.uc[0]) {
	  // gfs_get_from_below_intensive of the corrected velocities (coarse_init_kernel): the children
	  // of a coarse cell are 2 x 2 cells of this plane (k even: child ids 0-3) and of the plane
	  // before (k odd, ids 4-7), gathered from the lanes of the wave (two rows of the column: the
	  // first 32 lanes the lower j)
	  if (k & 1) {
#pragma unroll
	    for (int q = 0; q < 3; q++) low[q] = res[q];
	  }
	  else {
	    const int lane = tid & 63, lx = lane & 31;
	    const bool head = !(lane & 32) && !(lx & 1);
	    // id: bit 0 -> + x, bit 1 -> the lower j, (bit 2 -> the lower k: the plane before)
	    const int src[4] = { lx + 32, lx + 33, lx, lx + 1 };
#pragma unroll
	    for (int q = 0; q < 3; q++) {
	      double val = 0., sa = 0.;
#pragma unroll
	      for (int id = 0; id < 4; id++) { val += __shfl (res[q], src[id], 64)*1.; sa += 1.; }
#pragma unroll
	      for (int id = 0; id < 4; id++) { val += __shfl (low[q], src[id], 64)*1.; sa += 1.; }
	      if (head)
		K.uc[q][K.Lc.idx ((i + 1)/2, (j + 1)/2, k/2)] = val/sa;
	    }
	  }
	}
      }
#pragma unroll
      for (int q = 0; q < 3; q++)
	fpz[q] = Fp[q][2];
    }
    // ---- the states of plane p for the next iteration
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
      for (int d = 0; d < 3; d++)
	flo[q][d] = F[q].l[d];
      S_.FRx[q][ty][tx] = F[q].r[0];
      S_.FRy[q][ty][tx] = F[q].r[1];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// advect3_sweep2_kernel: the same sweep with every global load issued where nothing waits for it.
// The counters of advect3_sweep_kernel (profiles/r03_pmc_advect3.json: 75 % of the wave cycles in
// s_waitcnt / at barriers, the vector ALU 25 % busy) and its structure say where the time goes: three
// loads per plane are consumed right after they are issued -- the MAC velocities of the plane (written
// to LDS before the first barrier), gm (read by the flux stage), gc (read by the update) -- about 2 us
// of exposed latency each of the 8 us a plane takes.  Here un (p + 1) is loaded a whole iteration
// ahead, gm (p) and gc (p - 1) at the top of the iteration (consumed after the face values); to pay
// for those registers the left states of a plane wait in the thread's own LDS slots instead of
// registers, the x / y fluxes are re-read from LDS by the update, the right states go to LDS as soon
// as they exist (double-buffered by the parity of the plane), and the x / y fluxes of plane p - 1 are
// formed BEFORE the face values of plane p (only the z flux needs them).  Same expressions, same
// order per cell: bit-identical.
// ---------------------------------------------------------------------------------------------
struct Sweep2Lds {
  double V[3][SWY + 2][SWX + 2];
  double UNx[SWY][SWX + 1], UNy[SWY + 1][SWX];
  double FRx[2][3][SWY][SWX + 1], FRy[2][3][SWY + 1][SWX];   // right states of the planes p, p - 1 by parity
  double FPx[3][SWY][SWX + 1], FPy[3][SWY + 1][SWX];
  double GM[3][SWY + 1][SWX + 1];                      // gm of plane p - 1 (+ column and row after)
  double FL[3][3][SWY][SWX];                           // left states of plane p - 1: every thread its own slots
};

// MPI: a box with GfsBoundaryMpi sides (G.r[d] != nullptr: side d is one).  A ring cell beyond such a side is the
// ghost cell itself (its value, gm and the MAC velocities of its faces are those the neighbour box sent), and its
// state towards the column is what the neighbour computed for its own cell (boundary_face_values_kernel, one
// message per side) instead of the recomputed periodic image: G.r[d][q n^2 + f].  Planes 0 and n + 1 likewise
// along z (advect3_sweep2_kernel below).
template <bool MPI>
__device__ __forceinline__ int sweep2_plane (int k, int n, int sz, bool zlo, bool zhi)
{
  if (k < 1) k = MPI && zlo ? (k < 0 ? 0 : k) : k + n;
  else if (k > n) k = MPI && zhi ? n + 1 : k - n;
  return k*sz;
}

template <bool VL, bool SRC, bool MPI>
__device__ __forceinline__ void sweep2_ring_path (Sweep2Lds & S_, const Layout & L, const CPtr3 & v, const CPtr3 & un,
						  const CPtr3 & gm, double dt, const Visc3 & src3, int rid,
						  const GhostFv & G)
{
  const int n = L.n;
  const double rn = (double) n, rsize2 = (double) n/2.;
  const int sy0 = (int) L.sy, sz = (int) L.sz;
  const int kb = blockIdx.z*SWZ;
  const bool zlo = MPI && G.r[5] != nullptr, zhi = MPI && G.r[4] != nullptr;
  auto plane = [=] (int k) { return sweep2_plane<MPI> (k, n, sz, zlo, zhi); };
  int role = -1, hd = 0, hcol = 0, hslot = 0;
  const double * recv = nullptr;                     // the states of the ring cells beyond an MPI side, per plane n apart
  if (rid < 2*SWX) {
    role = rid < SWX ? 0 : 1; hd = 1; hslot = rid % SWX;
    int hj = blockIdx.y*SWY + (role ? SWY + 1 : 0);
    if (MPI && (hj < 1 || hj > n) && G.r[role ? 2 : 3])
      recv = G.r[role ? 2 : 3] + (blockIdx.x*SWX + hslot);          /* f = (x - 1) + n (z - 1) */
    else
      hj = hj < 1 ? hj + n : hj > n ? hj - n : hj;
    hcol = (int) L.idx (blockIdx.x*SWX + hslot + 1, hj, 0);
  }
  else if (rid < 2*SWX + 2*SWY) {
    role = rid < 2*SWX + SWY ? 2 : 3; hd = 0; hslot = (rid - 2*SWX) % SWY;
    int hi_ = blockIdx.x*SWX + (role == 3 ? SWX + 1 : 0);
    if (MPI && (hi_ < 1 || hi_ > n) && G.r[role == 3 ? 0 : 1])
      recv = G.r[role == 3 ? 0 : 1] + (blockIdx.y*SWY + hslot);    /* f = (y - 1) + n (z - 1) */
    else
      hi_ = hi_ < 1 ? hi_ + n : hi_ > n ? hi_ - n : hi_;
    hcol = (int) L.idx (hi_, blockIdx.y*SWY + hslot + 1, 0);
  }
  // a ghost row beyond a y side has no row beyond it: its y neighbours (never used: the state comes from recv)
  // are read from the row itself
  const int sy = MPI && recv && hd == 1 ? 0 : sy0;
  const size_t nf = (size_t) n*n;
  const bool ring = role >= 0, ring_minus = role == 0 || role == 2;
  // towards the column (a ring cell beyond an MPI side on the + side has nothing of its own to read there)
  const int hoff = MPI && recv && !ring_minus ? 0 : hd == 0 ? 1 : sy0;
  const int ry = role == 0 ? 0 : role == 1 ? SWY + 1 : hslot + 1;
  const int rx = role == 2 ? 0 : role == 3 ? SWX + 1 : hslot + 1;
  // everything is loaded one iteration before it is used (suffix n: for the next iteration)
  double hvm[3], hv0[3] = { 0., 0., 0. }, hvp[3] = { 0., 0., 0. }, hvn[3] = { 0., 0., 0. };
  double nbn[3][4];                                  // in-plane neighbours of the cell: x -, x +, y -, y +
  double huan[3] = { 0., 0., 0. }, hubn[2] = { 0., 0. };
  double hunb = 0., hunzb = 0., hlo[3] = { 0., 0., 0. }, hgmo[3], hgmn[3] = { 0., 0., 0. };
  double hgcn[3] = { 0., 0., 0. };                   // gm of the column's cell beside the ring cell
#pragma unroll
  for (int q = 0; q < 3; q++) {
    hgmo[q] = 0.;
#pragma unroll
    for (int e = 0; e < 4; e++) nbn[q][e] = 0.;
  }
  if (ring) {
    const int hc = hcol + plane (kb);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      hv0[q] = v.p[q][hcol + plane (kb - 1)];
      hvp[q] = v.p[q][hc];
      hvn[q] = v.p[q][hcol + plane (kb + 1)];
      nbn[q][0] = v.p[q][hc - 1];  nbn[q][1] = v.p[q][hc + 1];
      nbn[q][2] = v.p[q][hc - sy]; nbn[q][3] = v.p[q][hc + sy];
    }
    huan[0] = un.p[0][hc]; hubn[0] = un.p[0][hc - 1];
    huan[1] = un.p[1][hc]; hubn[1] = un.p[1][hc - sy];
    huan[2] = un.p[2][hc];
    hunzb = un.p[2][hcol + plane (kb - 1)];
  }
  for (int p = kb; p <= kb + SWZ + 1; p++) {
    const int zp = plane (p);
    Stencil7 HW[3];
    double hua[3], hub[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      hvm[q] = hv0[q]; hv0[q] = hvp[q]; hvp[q] = hvn[q];
      HW[q].v0 = hv0[q];
      HW[q].m[0] = nbn[q][0]; HW[q].p[0] = nbn[q][1];
      HW[q].m[1] = nbn[q][2]; HW[q].p[1] = nbn[q][3];
      HW[q].m[2] = hvm[q]; HW[q].p[2] = hvp[q];
      hgmo[q] = hgmn[q];                             // gm (p - 1) of the ring cell
    }
    hua[0] = huan[0]; hua[1] = huan[1]; hua[2] = huan[2];
    hub[0] = hubn[0]; hub[1] = hubn[1]; hub[2] = hunzb;
    const double hgc[3] = { hgcn[0], hgcn[1], hgcn[2] };
    if (ring) {
      // the loads of the next iteration (plane p + 1; gm of plane p)
      const int hc1 = hcol + plane (p + 1);
#pragma unroll
      for (int q = 0; q < 3; q++) {
	hvn[q] = v.p[q][hcol + plane (p + 2)];
	nbn[q][0] = v.p[q][hc1 - 1];  nbn[q][1] = v.p[q][hc1 + 1];
	nbn[q][2] = v.p[q][hc1 - sy]; nbn[q][3] = v.p[q][hc1 + sy];
	hgmn[q] = gm.p[q][hcol + zp];
	hgcn[q] = gm.p[q][hcol + zp + hoff];
      }
      huan[0] = un.p[0][hc1]; hubn[0] = un.p[0][hc1 - 1];
      huan[1] = un.p[1][hc1]; hubn[1] = un.p[1][hc1 - sy];
      huan[2] = un.p[2][hc1];
#pragma unroll
      for (int q = 0; q < 3; q++) {
	S_.V[q][ry][rx] = hv0[q];
	if (role == 3) S_.GM[q][hslot][SWX] = hgmo[q];
	if (role == 1) S_.GM[q][SWY][hslot] = hgmo[q];
      }
      if (role == 2) S_.UNx[hslot][0] = hua[0];
      if (role == 0) S_.UNy[0][hslot] = hua[1];
    }
    __syncthreads ();                                // (1)
    double hl[3] = { 0., 0., 0. };
    if (ring) {
      const AdvShared S = adv_shared_v (hua, hub, dt, rsize2);
#pragma unroll
      for (int q = 0; q < 3; q++) {
	FacePair f = hd == 0 ? adv_face_values_s<0, VL, SRC> (HW[q], S, dt, rsize2, src3.g[q]) :
	  adv_face_values_s<1, VL, SRC> (HW[q], S, dt, rsize2, src3.g[q]);
	if (MPI && recv) {
	  /* beyond an MPI side: the state the neighbour box sent (planes 0 and n + 1 feed nothing that is stored) */
	  const int zc = p < 1 ? 0 : p > n ? n - 1 : p - 1;
	  f.l = f.r = recv[q*nf + (size_t) n*zc];
	}
	hl[q] = f.l;
	if (role == 3) S_.FRx[p & 1][q][hslot][SWX] = f.r;
	if (role == 1) S_.FRy[p & 1][q][SWY][hslot] = f.r;
      }
      // the face between the cell and the column, plane p - 1
      if (p > kb && ring_minus) {
	const double ua = hunb;
	const double fu = 1.*ua*dt*rn;
#pragma unroll
	for (int q = 0; q < 3; q++) {
	  const double rs = role == 2 ? S_.FRx[(p - 1) & 1][q][hslot][0] : S_.FRy[(p - 1) & 1][q][0][hslot];
	  double f = fu;
	  f *= upwinded (ua, hlo[q], rs) - face_interp (hgmo[q], hgc[q])*dt/2.;
	  if (role == 2) S_.FPx[q][hslot][0] = f; else S_.FPy[q][0][hslot] = f;
	}
      }
    }
    __syncthreads ();                                // (2)
    if (ring) {
#pragma unroll
      for (int q = 0; q < 3; q++)
	hlo[q] = hl[q];
      hunb = hua[hd];
      hunzb = hua[2];
    }
  }
}

template <bool VL, bool SRC, bool CORR, bool MPI>
__global__ void __launch_bounds__(SWN + SW_RING)
advect3_sweep2_kernel (Layout L, CPtr3 v, Ptr3 out, CPtr3 un, CPtr3 gm, CPtr3 gc, double dt, Visc3 src3,
		       AdvCorr K, GhostFv G)
{
  __shared__ Sweep2Lds S_;
  const int tid = threadIdx.x;
  if (tid >= SWN) {
    sweep2_ring_path<VL, SRC, MPI> (S_, L, v, un, gm, dt, src3, tid - SWN, G);
    return;
  }
  const int tx = tid % SWX, ty = tid / SWX;
  const int n = L.n;
  const double rn = (double) n, rsize2 = (double) n/2.;
  const int sz = (int) L.sz;
  const int i = blockIdx.x*SWX + tx + 1, j = blockIdx.y*SWY + ty + 1;
  const int kb = blockIdx.z*SWZ;                     // output planes kb + 1 .. kb + SWZ
  const int col = (int) L.idx (i, j, 0);
  const bool zlo = MPI && G.r[5] != nullptr, zhi = MPI && G.r[4] != nullptr;
  auto plane = [=] (int k) { return sweep2_plane<MPI> (k, n, sz, zlo, zhi); };   // periodic image / ghost plane
  const size_t nf = (size_t) n*n;
  const int fpos = (i - 1) + n*(j - 1);              // face_pos (2, ...)

  double vm[3], v0[3], vp[3], vn[3];                 // planes p - 1, p, p + 1 and the load for p + 2
  double una[3], unb[3], unn[3];                     // un (p), un (p - 1), un (p + 1)
  double fpz[3] = { 0., 0., 0. };                    // fluxes through the z face below plane p - 1
  double gmo[3], gmc[3];                             // gm (p - 1), gm (p) of the own cell
  double low[3] = { 0., 0., 0. };                    // corrected values of the odd plane (coarse cells)
#pragma unroll
  for (int q = 0; q < 3; q++) {
    v0[q] = v.p[q][col + plane (kb - 1)];
    vp[q] = v.p[q][col + plane (kb)];
    vn[q] = v.p[q][col + plane (kb + 1)];
    gmo[q] = 0.;
    gmc[q] = gm.p[q][col + plane (kb - 1)];
#pragma unroll
    for (int d = 0; d < 3; d++) S_.FL[q][d][ty][tx] = 0.;
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
    una[d] = un.p[d][col + plane (kb - 1)];
    unn[d] = un.p[d][col + plane (kb)];
    unb[d] = 0.;
  }

  for (int p = kb; p <= kb + SWZ + 1; p++) {
    const int zp = plane (p);
    const int rp = p & 1, ro = rp ^ 1;
    // ---- shift the column registers; the loads (none is consumed before the face values are done,
    // un not before the next iteration)
    double gcv[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      vm[q] = v0[q]; v0[q] = vp[q]; vp[q] = vn[q];
      vn[q] = v.p[q][col + plane (p + 2)];
      gmo[q] = gmc[q];
      gmc[q] = gm.p[q][col + zp];
      gcv[q] = gc.p[q] ? gc.p[q][col + plane (p - 1)] : 0.;
    }
    const double unzb = una[2];                      // un_z (p - 1)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      unb[d] = una[d];
      una[d] = unn[d];
      unn[d] = un.p[d][col + plane (p + 1)];
    }
    // ---- the plane into LDS
#pragma unroll
    for (int q = 0; q < 3; q++) {
      S_.V[q][ty + 1][tx + 1] = v0[q];
      S_.GM[q][ty][tx] = gmo[q];
    }
    S_.UNx[ty][tx + 1] = una[0];
    S_.UNy[ty + 1][tx] = una[1];
    __syncthreads ();                                // (1)
    double flz[3] = { 0., 0., 0. };
    if (p > kb) {
      // ---- B (o), x and y: fluxes through the + faces of plane o = p - 1 from its left states (own LDS
      // slots), the right states of the neighbours (the other half of FR) and gm of the neighbours
      const double fux = 1.*unb[0]*dt*rn, fuy = 1.*unb[1]*dt*rn;
#pragma unroll
      for (int q = 0; q < 3; q++) {
	const double g0 = gmo[q];
	double f = fux;
	f *= upwinded (unb[0], S_.FL[q][0][ty][tx], S_.FRx[ro][q][ty][tx + 1]) - face_interp (g0, S_.GM[q][ty][tx + 1])*dt/2.;
	S_.FPx[q][ty][tx + 1] = f;
	f = fuy;
	f *= upwinded (unb[1], S_.FL[q][1][ty][tx], S_.FRy[ro][q][ty + 1][tx]) - face_interp (g0, S_.GM[q][ty + 1][tx])*dt/2.;
	S_.FPy[q][ty + 1][tx] = f;
	flz[q] = S_.FL[q][2][ty][tx];
      }
    }
    // ---- A (p): face values of plane p; the right states and the left states go to LDS at once
    double frz[3];
    {
      const double b[3] = { S_.UNx[ty][tx], S_.UNy[ty][tx], unzb };
      const AdvShared S = adv_shared_v (una, b, dt, rsize2);
#pragma unroll
      for (int q = 0; q < 3; q++) {
	Stencil7 W;
	W.v0 = v0[q];
	W.m[0] = S_.V[q][ty + 1][tx];     W.p[0] = S_.V[q][ty + 1][tx + 2];
	W.m[1] = S_.V[q][ty][tx + 1];     W.p[1] = S_.V[q][ty + 2][tx + 1];
	W.m[2] = vm[q];                   W.p[2] = vp[q];
	const SweepFv F = sweep_face_values<VL, SRC> (W, S, dt, rsize2, src3.g[q]);
	S_.FRx[rp][q][ty][tx] = F.r[0];
	S_.FRy[rp][q][ty][tx] = F.r[1];
	frz[q] = F.r[2];
#pragma unroll
	for (int d = 0; d < 3; d++)
	  S_.FL[q][d][ty][tx] = F.l[d];
	if (MPI) {
	  /* the planes beyond an MPI side along z: the states the neighbour boxes sent */
	  if (zlo && p == 0) S_.FL[q][2][ty][tx] = G.r[5][q*nf + fpos];
	  if (zhi && p == n + 1) frz[q] = G.r[4][q*nf + fpos];
	}
      }
    }
    double Fpz[3] = { 0., 0., 0. };
    if (p > kb) {
      // ---- B (o), z: the right state of plane p of the same thread
      const double fuz = 1.*unb[2]*dt*rn;
#pragma unroll
      for (int q = 0; q < 3; q++) {
	double f = fuz;
	f *= upwinded (unb[2], flz[q], frz[q]) - face_interp (gmo[q], gmc[q])*dt/2.;
	Fpz[q] = f;
      }
    }
    __syncthreads ();                                // (2) the fluxes of plane p - 1; plane p has been read
    if (p > kb) {
      const int o = p - 1, co = col + plane (o);
      if (p > kb + 1) {
	// ---- C (o): the gather in the reference's scatter order (flux_update_kernel), the update
	const int k = o;
	const unsigned J = n - j, K_ = n - k;
	const bool back_first = __ffs (~J) > __ffs (~K_);
	double res[3];
#pragma unroll
	for (int q = 0; q < 3; q++) {
	  const double Fmx = S_.FPx[q][ty][tx], Fmy = S_.FPy[q][ty][tx], Fmz = fpz[q];
	  const double Fpx = S_.FPx[q][ty][tx + 1], Fpy = S_.FPy[q][ty + 1][tx];
	  double acc = 0.;
	  if (i > 1)
	    acc += Fmx;
	  acc -= Fpx;
	  acc -= Fpy;
	  acc -= Fpz[q];
	  if (back_first) {
	    if (k > 1) acc += Fmz;
	    if (j > 1) acc += Fmy;
	  }
	  else {
	    if (j > 1) acc += Fmy;
	    if (k > 1) acc += Fmz;
	  }
	  if (i == 1)
	    acc += Fmx;
	  if (j == 1)
	    acc += Fmy;
	  if (k == 1)
	    acc += Fmz;
	  double val = vm[q];
	  val += acc/1.;
	  if (gc.p[q])
	    val -= gcv[q]*dt;
	  if (SRC && src3.g[q] != 0.) { /* gfs_domain_variable_centered_sources, src/source.c:62-108 */
	    double sum = 0;
	    sum += src3.g[q];
	    val += dt*sum;
	  }
	  if (CORR)          /* correct (src/timestep.c:486-496): u[c] -= g[c]*dt with dt = K.dt */
	    val = val - gcv[q]*K.dt;
	  out.p[q][co] = val;
	  res[q] = val;
	}
	if (CORR && K.uc[0]) {
	  // the first level of gfs_cell_coarse_init (see advect3_sweep_kernel)
	  if (k & 1) {
#pragma unroll
	    for (int q = 0; q < 3; q++) low[q] = res[q];
	  }
	  else {
	    const int lane = tid & 63, lx = lane & 31;
	    const bool head = !(lane & 32) && !(lx & 1);
	    const int src[4] = { lx + 32, lx + 33, lx, lx + 1 };
#pragma unroll
	    for (int q = 0; q < 3; q++) {
	      double val = 0., sa = 0.;
#pragma unroll
	      for (int id = 0; id < 4; id++) { val += __shfl (res[q], src[id], 64)*1.; sa += 1.; }
#pragma unroll
	      for (int id = 0; id < 4; id++) { val += __shfl (low[q], src[id], 64)*1.; sa += 1.; }
	      if (head)
		K.uc[q][K.Lc.idx ((i + 1)/2, (j + 1)/2, k/2)] = val/sa;
	    }
	  }
	}
      }
#pragma unroll
      for (int q = 0; q < 3; q++)
	fpz[q] = Fpz[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The predictor of the three MAC velocities as the same sweep along z (periodic boxes):
// gfs_face_advected_normal_velocity (src/advection.c:513-539) with the face values of u[d] along d
// (use_centered_velocity = TRUE), predict_un_tiled_kernel's expressions -- and, with div != nullptr,
// gfs_normal_divergence + scale_divergence of the MAC projection that follows (divergence_kernel) from
// the face velocities while they are in registers / LDS: the pass over un that reads them back is
// saved.  Pipeline as in advect3_sweep_kernel: A (p) the left / right states of u[d] along d of
// plane p; B (p - 1) the face velocities of the + faces of plane p - 1; C (p - 1) the divergence.
// ---------------------------------------------------------------------------------------------
struct PredLds {
  double V[3][SWY + 2][SWX + 2];
  double FRx[2][SWY][SWX + 1], FRy[2][SWY + 1][SWX];     // right states of u_x along x, u_y along y, by parity
  double UOx[SWY][SWX + 1], UOy[SWY + 1][SWX];           // face velocities of plane p - 1 (+ ring before)
};

// face values of v = u[D] along D in a cell whose stencil W of u[D] and centred velocities uc are
// loaded: face_values_dir<3, D, true, VL, VS>
template <int D, bool VL, bool VS>
__device__ __forceinline__ FacePair cen_face_values_s (const Stencil7 & W, const double uc[3], double dt,
						       int n, double visc, double gsrc)
{
  const double rsize = (double) n, rsize2 = (double) n/2.;
  constexpr int A = D == 0 ? 1 : 0, B = D == 2 ? 1 : 2;
  const double ta = adv_transverse_v (W.m[A], W.p[A], W.v0, uc[A], dt, rsize2);
  const double tb = adv_transverse_v (W.m[B], W.p[B], W.v0, uc[B], dt, rsize2);
  const double unorm = dt*uc[D]*rsize;
  const double v0 = W.v0, v1 = W.m[D], v2 = W.p[D];
  const double g = VL ? van_leer_gradient (v0, v1, v2) : center_gradient (v0, v1, v2);
  const double vl = v0 + GMIN ((1. - unorm)/2., 0.5)*g;
  const double vr = v0 + GMAX ((- 1. - unorm)/2., -0.5)*g;
  double msrc = 0.;
  if (VS && visc != 0.) {
    /* source_diffusion_value, src/source.c:1105-1144 */
    const double rh2 = (double) n*(double) n;
    double ga = 0., gb = 0.;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      ga += visc*1.; gb += visc*W.p[cc];
      ga += visc*1.; gb += visc*W.m[cc];
    }
    msrc = 0. + 1.*(gb - ga*v0)*rh2;
  }
  if (VS) msrc += gsrc;
  const double src = dt*msrc/2.;
  double dv = ta;
  dv += tb;
  FacePair f;
  f.l = vl + src - dv;
  f.r = vr + src - dv;
  return f;
}

template <bool VL, bool VS, bool MPI>
__device__ __forceinline__ void pred_ring_path (PredLds & S_, const Layout & L, const CPtr3 & u, double dt,
						const Visc3 & visc, int rid, const Ptr3 & un, const GhostFv & G)
{
  const int n = L.n;
  const int sy0 = (int) L.sy, sz = (int) L.sz;
  const int kb = blockIdx.z*SWZ;
  const bool zlo = MPI && G.r[5] != nullptr, zhi = MPI && G.r[4] != nullptr;
  auto plane = [=] (int k) { return sweep2_plane<MPI> (k, n, sz, zlo, zhi); };
  int role = -1, hd = 0, hcol = 0, hslot = 0;
  // MPI: the state of a ring cell beyond an MPI side is the one the neighbour box sent (one variable per
  // direction: u[d] along d), the cell itself is the ghost cell; the face between it and the column on the low
  // side of the box is computed here (the periodic box copies it from the high side)
  const double * recv = nullptr;
  if (rid < 2*SWX) {
    role = rid < SWX ? 0 : 1; hd = 1; hslot = rid % SWX;
    int hj = blockIdx.y*SWY + (role ? SWY + 1 : 0);
    if (MPI && (hj < 1 || hj > n) && G.r[role ? 2 : 3])
      recv = G.r[role ? 2 : 3] + (blockIdx.x*SWX + hslot);
    else
      hj = hj < 1 ? hj + n : hj > n ? hj - n : hj;
    hcol = (int) L.idx (blockIdx.x*SWX + hslot + 1, hj, 0);
  }
  else if (rid < 2*SWX + 2*SWY) {
    role = rid < 2*SWX + SWY ? 2 : 3; hd = 0; hslot = (rid - 2*SWX) % SWY;
    int hi_ = blockIdx.x*SWX + (role == 3 ? SWX + 1 : 0);
    if (MPI && (hi_ < 1 || hi_ > n) && G.r[role == 3 ? 0 : 1])
      recv = G.r[role == 3 ? 0 : 1] + (blockIdx.y*SWY + hslot);
    else
      hi_ = hi_ < 1 ? hi_ + n : hi_ > n ? hi_ - n : hi_;
    hcol = (int) L.idx (hi_, blockIdx.y*SWY + hslot + 1, 0);
  }
  const bool ring = role >= 0, ring_minus = role == 0 || role == 2;
  const int sy = MPI && recv && hd == 1 ? 0 : sy0;          /* no row beyond a ghost row: never used then */
  const int hoff = MPI && recv && !ring_minus ? 0 : hd == 0 ? 1 : sy0;
  const int ry = role == 0 ? 0 : role == 1 ? SWY + 1 : hslot + 1;
  const int rx = role == 2 ? 0 : role == 3 ? SWX + 1 : hslot + 1;
  // the column of the ring cell: the three components (the ring of V), of which u[hd] is advected
  double hvm[3], hv0[3] = { 0., 0., 0. }, hvp[3] = { 0., 0., 0. }, hvn[3] = { 0., 0., 0. };
  double hlo = 0., hnb = 0., hnbn = 0.;        // left state of plane p - 1; u[hd] of the column's cell beside, planes p - 1 / p
  double nbn[4] = { 0., 0., 0., 0. };          // in-plane neighbours of u[hd] at the cell, loaded one iteration ahead
  if (ring) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      hv0[q] = u.p[q][hcol + plane (kb - 1)];
      hvp[q] = u.p[q][hcol + plane (kb)];
      hvn[q] = u.p[q][hcol + plane (kb + 1)];
    }
    hnbn = u.p[hd][hcol + hoff + plane (kb - 1)];
    const double * __restrict__ const w = u.p[hd];
    const int hc = hcol + plane (kb);
    nbn[0] = w[hc - 1];  nbn[1] = w[hc + 1];
    nbn[2] = w[hc - sy]; nbn[3] = w[hc + sy];
  }
  for (int p = kb; p <= kb + SWZ + 1; p++) {
    const int zp = plane (p);
    Stencil7 HW;
    HW.m[0] = HW.p[0] = HW.m[1] = HW.p[1] = 0.;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      hvm[q] = hv0[q]; hv0[q] = hvp[q]; hvp[q] = hvn[q];
    }
    HW.v0 = hv0[hd]; HW.m[2] = hvm[hd]; HW.p[2] = hvp[hd];
    HW.m[0] = nbn[0]; HW.p[0] = nbn[1]; HW.m[1] = nbn[2]; HW.p[1] = nbn[3];
    hnb = hnbn;
    if (ring) {
      const int hc1 = hcol + plane (p + 1);
#pragma unroll
      for (int q = 0; q < 3; q++)
	hvn[q] = u.p[q][hcol + plane (p + 2)];
      const double * __restrict__ const w = u.p[hd];
      nbn[0] = w[hc1 - 1];  nbn[1] = w[hc1 + 1];     /* plane p + 1: for the next iteration */
      nbn[2] = w[hc1 - sy]; nbn[3] = w[hc1 + sy];
      hnbn = w[hcol + hoff + zp];                 /* plane p: the face of the next iteration */
#pragma unroll
      for (int q = 0; q < 3; q++)
	S_.V[q][ry][rx] = hv0[q];
    }
    __syncthreads ();                                // (1)
    double hl = 0., hr = 0.;
    if (ring) {
      const FacePair f = hd == 0 ? cen_face_values_s<0, VL, VS> (HW, hv0, dt, n, visc.d[0], visc.g[0]) :
	cen_face_values_s<1, VL, VS> (HW, hv0, dt, n, visc.d[1], visc.g[1]);
      hl = f.l; hr = f.r;
      if (MPI && recv) {
	const int zc = p < 1 ? 0 : p > n ? n - 1 : p - 1;
	hl = hr = recv[(size_t) n*zc];
      }
      if (role == 3) S_.FRx[p & 1][hslot][SWX] = hr;
      if (role == 1) S_.FRy[p & 1][SWY][hslot] = hr;
      // the face between the cell and the column, plane p - 1 (for the divergence of the column's cells)
      if (p > kb && ring_minus) {
	const double rs = role == 2 ? S_.FRx[(p - 1) & 1][hslot][0] : S_.FRy[(p - 1) & 1][0][hslot];
	const double s0 = face_interp (hvm[hd], hnb);
	const double val = upwinded (s0, hlo, rs);
	if (role == 2) S_.UOx[hslot][0] = val; else S_.UOy[0][hslot] = val;
	if (MPI && recv && p - 1 >= 1 && p - 1 <= n)      /* the face on the low side of the box */
	  un.p[hd][hcol + plane (p - 1)] = val;
      }
    }
    __syncthreads ();                                // (2)
    hlo = hl;
  }
}

template <bool VL, bool VS, bool DIV, bool MPI>
__global__ void __launch_bounds__(SWN + SW_RING)
predict_un_sweep_kernel (Layout L, CPtr3 u, Ptr3 un, double dt, Visc3 visc, double * __restrict__ div,
			 double div_dt, GhostFv G)
{
  __shared__ PredLds S_;
  const int tid = threadIdx.x;
  if (tid >= SWN) {
    pred_ring_path<VL, VS, MPI> (S_, L, u, dt, visc, tid - SWN, un, G);
    return;
  }
  const int tx = tid % SWX, ty = tid / SWX;
  const int n = L.n;
  const int sy = (int) L.sy, sz = (int) L.sz;
  const int i = blockIdx.x*SWX + tx + 1, j = blockIdx.y*SWY + ty + 1;
  const int kb = blockIdx.z*SWZ;
  const int col = (int) L.idx (i, j, 0);
  const bool zlo = MPI && G.r[5] != nullptr, zhi = MPI && G.r[4] != nullptr;
  const bool xmpi = MPI && G.r[1] != nullptr, ympi = MPI && G.r[3] != nullptr;
  auto plane = [=] (int k) { return sweep2_plane<MPI> (k, n, sz, zlo, zhi); };
  const int fpos = (i - 1) + n*(j - 1);
  double vm[3], v0[3], vp[3], vn[3];
  double flo[3] = { 0., 0., 0. };                    // left states of plane p - 1 (u[d] along d)
  double uxp = 0., uyp = 0.;                         // u_x (c + 1), u_y (c + sy) of plane p - 1
  double unzo = 0.;                                  // un_z of the plane before p - 1
#pragma unroll
  for (int q = 0; q < 3; q++) {
    v0[q] = u.p[q][col + plane (kb - 1)];
    vp[q] = u.p[q][col + plane (kb)];
    vn[q] = u.p[q][col + plane (kb + 1)];
  }
  for (int p = kb; p <= kb + SWZ + 1; p++) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      vm[q] = v0[q]; v0[q] = vp[q]; vp[q] = vn[q];
      vn[q] = u.p[q][col + plane (p + 2)];
      S_.V[q][ty + 1][tx + 1] = v0[q];
    }
    __syncthreads ();                                // (1)
    const int rp = p & 1, ro = rp ^ 1;
    FacePair F[3];
    double uxpn, uypn;
    {
      Stencil7 W;
      // u_x along x
      W.v0 = v0[0];
      W.m[0] = S_.V[0][ty + 1][tx];  W.p[0] = S_.V[0][ty + 1][tx + 2];
      W.m[1] = S_.V[0][ty][tx + 1];  W.p[1] = S_.V[0][ty + 2][tx + 1];
      W.m[2] = vm[0];                W.p[2] = vp[0];
      uxpn = W.p[0];
      F[0] = cen_face_values_s<0, VL, VS> (W, v0, dt, n, visc.d[0], visc.g[0]);
      S_.FRx[rp][ty][tx] = F[0].r;
      // u_y along y
      W.v0 = v0[1];
      W.m[0] = S_.V[1][ty + 1][tx];  W.p[0] = S_.V[1][ty + 1][tx + 2];
      W.m[1] = S_.V[1][ty][tx + 1];  W.p[1] = S_.V[1][ty + 2][tx + 1];
      W.m[2] = vm[1];                W.p[2] = vp[1];
      uypn = W.p[1];
      F[1] = cen_face_values_s<1, VL, VS> (W, v0, dt, n, visc.d[1], visc.g[1]);
      S_.FRy[rp][ty][tx] = F[1].r;
      // u_z along z
      W.v0 = v0[2];
      W.m[0] = S_.V[2][ty + 1][tx];  W.p[0] = S_.V[2][ty + 1][tx + 2];
      W.m[1] = S_.V[2][ty][tx + 1];  W.p[1] = S_.V[2][ty + 2][tx + 1];
      W.m[2] = vm[2];                W.p[2] = vp[2];
      F[2] = cen_face_values_s<2, VL, VS> (W, v0, dt, n, visc.d[2], visc.g[2]);
      if (MPI) {
	/* the planes beyond an MPI side along z: the states the neighbour boxes sent */
	if (zlo && p == 0) F[2].l = G.r[5][fpos];
	if (zhi && p == n + 1) F[2].r = G.r[4][fpos];
      }
    }
    double unv[3] = { 0., 0., 0. };
    if (p > kb) {
      // ---- B (o): the face velocities of the + faces of plane o = p - 1
      const int o = p - 1, co = col + plane (o);
      unv[0] = upwinded (face_interp (vm[0], uxp), flo[0], S_.FRx[ro][ty][tx + 1]);
      unv[1] = upwinded (face_interp (vm[1], uyp), flo[1], S_.FRy[ro][ty + 1][tx]);
      unv[2] = upwinded (face_interp (vm[2], v0[2]), flo[2], F[2].r);
      if (o >= 1 && o <= n) {
	un.p[0][co] = unv[0];
	un.p[1][co] = unv[1];
	un.p[2][co] = unv[2];
	// the face on the low side of the box is the periodic image of the one on the high side (beyond an MPI
	// side it is the face of the ghost cell: the ring lanes / the plane 0 compute it)
	if (i == n && !xmpi) un.p[0][co - n] = unv[0];
	if (j == n && !ympi) un.p[1][co - n*sy] = unv[1];
	if (o == n && !zlo) un.p[2][co - n*sz] = unv[2];
      }
      else if (MPI && zlo && o == 0)
	un.p[2][co] = unv[2];
      if (DIV) {
	S_.UOx[ty][tx + 1] = unv[0];
	S_.UOy[ty + 1][tx] = unv[1];
      }
    }
    __syncthreads ();                                // (2)
    if (p > kb) {
      if (DIV && p > kb + 1) {
	// ---- C (o): gfs_normal_divergence + scale_divergence (divergence_kernel)
	const int co = col + plane (p - 1);
	const double h = 1./n;
	double d_ = 0.;
	d_ += 1.*unv[0]*1.;
	d_ += -1.*S_.UOx[ty][tx]*1.;
	d_ += 1.*unv[1]*1.;
	d_ += -1.*S_.UOy[ty][tx]*1.;
	d_ += 1.*unv[2]*1.;
	d_ += -1.*unzo*1.;
	const double w_ = d_*h;
	div[co] = w_/div_dt;
      }
      unzo = unv[2];
    }
#pragma unroll
    for (int d = 0; d < 3; d++)
      flo[d] = F[d].l;
    uxp = uxpn; uyp = uypn;
  }
}

// ---------------------------------------------------------------------------------------------
// K19: minimum_mac_cfl / minimum_cfl, src/domain.c:2824-2923: min of (h/|un|)^2 and (h/|u|)^2
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_min (double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v = fmin (v, __shfl_down (v, o, 64));
  return v;
}

template <int DIM>
__global__ void __launch_bounds__(256)
cfl_partial_kernel (Layout L, CPtr3 u, CPtr3 un, Visc3 visc, double * __restrict__ partial)
{
  const int n = L.n;
  const double length = 1./n;
  const long r = n + 1;
  const long next = DIM == 3 ? r*r*r : r*r;
  const long off[3] = { 1, L.sy, L.sz };
  double m = DBL_MAX;
  for (long q = (long) blockIdx.x*blockDim.x + threadIdx.x; q < next;
       q += (long) gridDim.x*blockDim.x) {
    int i = q % r, j = (q / r) % r, k = DIM == 3 ? q / (r*r) : 0;
    long c = L.idx (i, j, k);
#pragma unroll
    for (int cc = 0; cc < DIM; cc++)
      if (face_valid<DIM> (n, cc, i, j, k)) {
	double unf = un.p[cc][c];
	if (unf != 0.) {
	  double cflu = length/fabs (unf);
	  m = fmin (m, cflu*cflu);
	}
      }
    if (i >= 1 && j >= 1 && (DIM == 2 || k >= 1)) {
#pragma unroll
      for (int cc = 0; cc < DIM; cc++) {
	double uv = u.p[cc][c];
	if (uv != 0.) {
	  double cflu = length/fabs (1.*uv);
	  m = fmin (m, cflu*cflu);
	}
	if ((visc.d[cc] != 0. || visc.g[cc] != 0.) && i <= n && j <= n && (DIM == 2 || k <= n)) {
	  /* p->v[c]->sources: acceleration time scale, src/domain.c:2893-2901 (gfs_variable_mac_source:
	     the explicit diffusion term, then the intensity of a GfsSource) */
	  double g = 0.;
	  if (visc.d[cc] != 0.) g += source_diffusion_value<DIM> (u.p[cc], c, off, visc.d[cc], n);
	  if (visc.g[cc] != 0.) g += visc.g[cc];
	  if (g != 0.) {
	    double cflg = 2.*length/fabs (1.*g);
	    m = fmin (m, cflg);
	  }
	}
      }
    }
  }
  __shared__ double sh[4];
  m = wave_min (m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads ();
  if (threadIdx.x == 0) {
    double rmin = sh[0];
    for (int q = 1; q < (int) (blockDim.x >> 6); q++) rmin = fmin (rmin, sh[q]);
    partial[blockIdx.x] = rmin;
  }
}

__global__ void __launch_bounds__(256)
min_final_kernel (const double * __restrict__ partial, int nblocks, double * __restrict__ out)
{
  double m = DBL_MAX;
  for (int q = threadIdx.x; q < nblocks; q += blockDim.x)
    m = fmin (m, partial[q]);
  __shared__ double sh[4];
  m = wave_min (m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads ();
  if (threadIdx.x == 0)
    out[0] = fmin (fmin (sh[0], sh[1]), fmin (sh[2], sh[3]));
}

// K18: gfs_get_from_below_intensive, src/fluid.c:1843-1864 (cell fractions = 1.)
struct CoarseArgs { double * vc[8]; const double * vf[8]; int nf; };

template <int DIM>
__global__ void __launch_bounds__(256)
coarse_init_kernel (Layout Lc, Layout Lf, CoarseArgs A)
{
  CELL_PROLOGUE (Lc);
  for (int f = 0; f < A.nf; f++) {        // every variable of the domain (src/adaptive.c:43-58)
    double val = 0., sa = 0.;
#pragma unroll
    for (int id = 0; id < (1 << DIM); id++) {
      int ci = 2*i - 1 + (id & 1);
      int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
      int ck = DIM == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
      double a = 1.;
      val += A.vf[f][Lf.idx (ci, cj, ck)]*a;
      sa += a;
    }
    A.vc[f][c] = val/sa;
  }
}

// derived variable Divergence: gfs_divergence, src/fluid.c:2357-2376
template <int DIM>
__global__ void __launch_bounds__(256)
velocity_divergence_kernel (Layout L, CPtr3 u, double * __restrict__ out)
{
  CELL_PROLOGUE (L);
  const long off[3] = { 1, L.sy, L.sz };
  double h = 1./L.n;
  double div = 0.;
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    div += 1.*1.*face_interp (u.p[cc][c], u.p[cc][c + off[cc]]);
    div += 1.*-1.*face_interp (u.p[cc][c], u.p[cc][c - off[cc]]);
  }
  out[c] = div/(1.*h);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define DISPATCH(dom, kern, grid, block, ...)				\
  do {									\
    if ((dom)->dim == 3)						\
      hipLaunchKernelGGL (kern<3>, grid, block, 0, (dom)->stream, __VA_ARGS__); \
    else								\
      hipLaunchKernelGGL (kern<2>, grid, block, 0, (dom)->stream, __VA_ARGS__); \
    GFSHIP_HIP (hipGetLastError ());					\
  } while (0)

static CPtr3 c3 (double * const p[3]) { CPtr3 r; for (int q = 0; q < 3; q++) r.p[q] = p[q]; return r; }
static Ptr3  m3 (double * const p[3]) { Ptr3 r; for (int q = 0; q < 3; q++) r.p[q] = p[q]; return r; }
static CPtr6 c6 (double * const p[6]) { CPtr6 r; for (int q = 0; q < 6; q++) r.p[q] = p[q]; return r; }
static Ptr6  m6 (double * const p[6]) { Ptr6 r; for (int q = 0; q < 6; q++) r.p[q] = p[q]; return r; }

int launch_face_interp_un (gfship_domain * dom, double * const u[3], double * const un[3])
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  ext_grid (L, &grid, &block);
  DISPATCH (dom, face_interp_un_kernel, grid, block, L, c3 (u), m3 (un));
  return GFSHIP_OK;
}

int launch_divergence (gfship_domain * dom, double * const un[3], double * div, double dt)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  DISPATCH (dom, divergence_kernel, grid, block, L, c3 (un), div, dt);
  return GFSHIP_OK;
}

int launch_correct_un (gfship_domain * dom, const double * p, double * const un[3], double dt)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  ext_grid (L, &grid, &block);
  DISPATCH (dom, correct_un_kernel, grid, block, L, p, m3 (un), dt);
  return GFSHIP_OK;
}

int launch_centered_gradient (gfship_domain * dom, const double * p, double * const g[3])
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  DISPATCH (dom, centered_gradient_kernel, grid, block, L, p, m3 (g));
  return GFSHIP_OK;
}

int launch_correct_centered (gfship_domain * dom, double * const u[3], double * const g[3], double dt)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  DISPATCH (dom, correct_centered_kernel, grid, block, L, m3 (u), c3 (g), dt);
  return GFSHIP_OK;
}

int launch_correct_centered_coarse (gfship_domain * dom, double * const u[3], double * const g[3],
				    double dt, double * const uc[3])
{
  const Layout & L = dom->lay[dom->depth], & Lc = dom->lay[dom->depth - 1];
  dim3 grid, block;
  cell_grid (Lc, &grid, &block);
  hipLaunchKernelGGL (correct_centered_coarse_kernel, grid, block, 0, dom->stream, L, Lc, m3 (u), c3 (g), dt,
		      m3 (uc));
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int launch_advected_face_values (gfship_domain * dom, const double * v, double * const u[3],
				 double * const un[3], double dt, int use_centered, int gradient,
				 double * const fv[6], int cmask, double visc, double gsrc)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  DISPATCH (dom, advected_face_values_kernel, grid, block, L, v, c3 (u), c3 (un), dt,
	    use_centered, gradient, m6 (fv), cmask, visc, gsrc);
  return GFSHIP_OK;
}

int launch_face_bc (gfship_domain * dom, Field * v, double * const fv[6], int cmask)
{
  const Layout & L = dom->lay[dom->depth];
  BcDesc bc;
  for (int d = 0; d < 6; d++) {
    bc.side[d] = dom->side[d];
    bc.type[d] = v->bc[d];
    bc.val[d] = v->bcval[d];
  }
  bc.component = v->component;
  bc.homogeneous = 0;
  int nface = dom->dim == 3 ? L.n*L.n : L.n;
  int block = nface >= 256 ? 256 : 64;
  dim3 grid ((nface + block - 1)/block, 2*dom->dim);
  DISPATCH (dom, face_bc_kernel, grid, dim3 (block), L, bc, (const double *) v->lev[dom->depth],
	    m6 (fv), cmask);
  /* GfsBoundaryMpi sides: ghost face values fv[e] beyond side e^1 come from the neighbour box */
  if (dom->has_external)
    for (int e = 0; e < 2*dom->dim; e++)
      if (dom->side[e ^ 1] == GFSHIP_SIDE_EXTERNAL && (cmask & (1 << (e/2)))) {
	int r = call_exchange (dom, fv[e], dom->depth, 1 + e);
	if (r) return r;
      }
  return GFSHIP_OK;
}

int launch_predict_un (gfship_domain * dom, int cc, const double * uc, double * const fv[6],
		       double * unc)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  ext_grid (L, &grid, &block);
  DISPATCH (dom, predict_un_kernel, grid, block, L, cc, uc, (const double *) fv[2*cc],
	    (const double *) fv[2*cc + 1], unc);
  return GFSHIP_OK;
}

int launch_flux_update (gfship_domain * dom, bool velocity, double * v, double * const un[3],
			double * const fv[6], const double * gm, const double * gc, double dt, double gsrc)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  if (dom->dim == 3) {
    if (velocity)
      hipLaunchKernelGGL ((flux_update_kernel<3, true>), grid, block, 0, dom->stream,
			  L, v, c3 (un), c6 (fv), gm, gc, dt, gsrc);
    else
      hipLaunchKernelGGL ((flux_update_kernel<3, false>), grid, block, 0, dom->stream,
			  L, v, c3 (un), c6 (fv), gm, gc, dt, gsrc);
  }
  else {
    if (velocity)
      hipLaunchKernelGGL ((flux_update_kernel<2, true>), grid, block, 0, dom->stream,
			  L, v, c3 (un), c6 (fv), gm, gc, dt, gsrc);
    else
      hipLaunchKernelGGL ((flux_update_kernel<2, false>), grid, block, 0, dom->stream,
			  L, v, c3 (un), c6 (fv), gm, gc, dt, gsrc);
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// grid of the kernels above: i = 1..n in blocks along x (the thread of i = 1 also does i = 0),
// j and k = 0..n
static inline void ext1_grid (const Layout & L, dim3 * grid, dim3 * block)
{
  int b = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
  *block = dim3 (b);
  *grid = dim3 ((L.n + b - 1)/b, L.n + 1, L.dim == 3 ? L.n + 1 : 1);
}

int launch_project_correct (gfship_domain * dom, const double * p, double * const un[3],
			    double * const g[3], double * const u[3] /* or nullptr */, double dt,
			    bool want_max)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  ext1_grid (L, &grid, &block);
  double * pm = nullptr;
  if (want_max) {
    const size_t nb = 1024;
    if (!dom->cfl_partial) {
      GFSHIP_HIP (hipMalloc ((void **) &dom->cfl_partial, nb*sizeof (double)));
      GFSHIP_HIP (hipMemsetAsync (dom->cfl_partial, 0, nb*sizeof (double), dom->stream));
      dom->cfl_nblocks = nb;
    }
    dom->cfl_used = nb;
    if (dom->cfl_dirty)     /* maxima of an earlier projection that nobody consumed */
      GFSHIP_HIP (hipMemsetAsync (dom->cfl_partial, 0, nb*sizeof (double), dom->stream));
    dom->cfl_dirty = true;
    pm = dom->cfl_partial;
  }
  Ptr3 none = { { nullptr, nullptr, nullptr } };
  static const bool pairs = getenv ("GFSHIP_PC_SCALAR") == nullptr;
  if (dom->dim == 3 && !u && !pm && pairs && L.n >= 64 && L.n % 2 == 0) {
    /* the MAC projection: two cells per thread, 16-byte accesses */
    const int b = 128, half = L.n/2;
    hipLaunchKernelGGL (project_correct2_kernel, dim3 ((half + b - 1)/b, L.n + 1, L.n + 1), dim3 (b), 0, dom->stream,
			L, p, m3 (un), m3 (g), dt);
    GFSHIP_HIP (hipGetLastError ());
    return GFSHIP_OK;
  }
  if (dom->dim == 3) {
    if (u) hipLaunchKernelGGL ((project_correct_kernel<3, true>), grid, block, 0, dom->stream, L, p, m3 (un), m3 (g), m3 (u), dt, pm);
    else   hipLaunchKernelGGL ((project_correct_kernel<3, false>), grid, block, 0, dom->stream, L, p, m3 (un), m3 (g), none, dt, pm);
  }
  else {
    if (u) hipLaunchKernelGGL ((project_correct_kernel<2, true>), grid, block, 0, dom->stream, L, p, m3 (un), m3 (g), m3 (u), dt, pm);
    else   hipLaunchKernelGGL ((project_correct_kernel<2, false>), grid, block, 0, dom->stream, L, p, m3 (un), m3 (g), none, dt, pm);
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// gfs_correct_normal_velocities + gfs_scale_gradients (+ gfs_correct_centered_velocities) with the face
// weights of the domain (gfship_poisson_coefficients_alpha); un == nullptr: gfs_update_gradients
int launch_project_correct_weighted (gfship_domain * dom, const double * p, double * const un[3] /* or nullptr */,
				     double * const g[3], double * const u[3] /* or nullptr */, double dt)
{
  const int Lv = dom->depth;
  const Layout & L = dom->lay[Lv];
  dim3 grid, block;
  ext1_grid (L, &grid, &block);
  W6t wf;
  for (int d = 0; d < 6; d++)
    wf.p[d] = d < 2*dom->dim ? dom->fields[dom->wf[d]].lev[Lv] : nullptr;
  Ptr3 none = { { nullptr, nullptr, nullptr } };
  if (dom->dim == 3)
    hipLaunchKernelGGL ((project_correct_weighted_kernel<3>), grid, block, 0, dom->stream, L, p,
			un ? m3 (un) : none, m3 (g), u ? m3 (u) : none, dt, wf);
  else
    hipLaunchKernelGGL ((project_correct_weighted_kernel<2>), grid, block, 0, dom->stream, L, p,
			un ? m3 (un) : none, m3 (g), u ? m3 (u) : none, dt, wf);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// the CFL time scale from the maxima left by launch_project_correct (want_max)
int launch_cfl_from_max (gfship_domain * dom, double * cfl2)
{
  const Layout & L = dom->lay[dom->depth];
  double * result = dom->h_pinned;      /* host memory mapped on the device: no copy kernel */
  hipLaunchKernelGGL (cfl_from_max_kernel, dim3 (1), dim3 (256), 0, dom->stream, dom->cfl_partial,
		      (int) dom->cfl_used, 1./L.n, result);
  dom->cfl_dirty = false;
  GFSHIP_HIP (hipGetLastError ());
  GFSHIP_HIP (stream_wait_spin (dom->stream));
  *cfl2 = dom->h_pinned[0];
  return call_reduce (dom, cfl2, 1, 2);
}

int launch_face_interp_div (gfship_domain * dom, double * const u[3], double * const un[3],
			    double * div, double dt)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  ext1_grid (L, &grid, &block);
  if (un) {
    if (dom->dim == 3) hipLaunchKernelGGL ((face_interp_div_kernel<3, true>), grid, block, 0, dom->stream, L, c3 (u), m3 (un), div, dt);
    else hipLaunchKernelGGL ((face_interp_div_kernel<2, true>), grid, block, 0, dom->stream, L, c3 (u), m3 (un), div, dt);
  }
  else {
    Ptr3 none = { { nullptr, nullptr, nullptr } };
    static const bool pairs = getenv ("GFSHIP_PC_SCALAR") == nullptr;
    if (dom->dim == 3 && pairs && L.n >= 64 && L.n % 2 == 0) {
      const int b = 128, half = L.n/2;
      hipLaunchKernelGGL (face_interp_div2_kernel, dim3 ((half + b - 1)/b, L.n, L.n), dim3 (b), 0, dom->stream, L, c3 (u), div, dt);
    }
    else if (dom->dim == 3) hipLaunchKernelGGL ((face_interp_div_kernel<3, false>), grid, block, 0, dom->stream, L, c3 (u), none, div, dt);
    else hipLaunchKernelGGL ((face_interp_div_kernel<2, false>), grid, block, 0, dom->stream, L, c3 (u), none, div, dt);
  }
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// the update of the approximate projection without the MAC velocities (project_correct_lazy_kernel):
// g and the corrected centred velocities uo from p and the uncorrected u; the maxima for the CFL
// condition are always left behind
int launch_project_correct_lazy (gfship_domain * dom, const double * p, double * const u[3],
				 double * const g[3], double * const uo[3], double dt)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  ext1_grid (L, &grid, &block);
  const size_t nb = 1024;
  if (!dom->cfl_partial) {
    GFSHIP_HIP (hipMalloc ((void **) &dom->cfl_partial, nb*sizeof (double)));
    GFSHIP_HIP (hipMemsetAsync (dom->cfl_partial, 0, nb*sizeof (double), dom->stream));
    dom->cfl_nblocks = nb;
  }
  dom->cfl_used = nb;
  if (dom->cfl_dirty)
    GFSHIP_HIP (hipMemsetAsync (dom->cfl_partial, 0, nb*sizeof (double), dom->stream));
  dom->cfl_dirty = true;
  static const bool pairs = getenv ("GFSHIP_PC_SCALAR") == nullptr;
  if (dom->dim == 3 && pairs && L.n >= 64 && L.n % 2 == 0) {
    /* two cells per thread, 16-byte accesses */
    const int b = 128, half = L.n/2;
    hipLaunchKernelGGL (project_correct_lazy2_kernel, dim3 ((half + b - 1)/b, L.n + 1, L.n + 1), dim3 (b), 0, dom->stream,
			L, p, c3 (u), m3 (g), m3 (uo), dt, dom->cfl_partial);
  }
  else if (dom->dim == 3)
    hipLaunchKernelGGL ((project_correct_lazy_kernel<3>), grid, block, 0, dom->stream, L, p, c3 (u), m3 (g), m3 (uo), dt, dom->cfl_partial);
  else
    hipLaunchKernelGGL ((project_correct_lazy_kernel<2>), grid, block, 0, dom->stream, L, p, c3 (u), m3 (g), m3 (uo), dt, dom->cfl_partial);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

bool godunov_fused_supported (const gfship_domain * dom)
{
  if (dom->dim != 3 || dom->no_fused_godunov) return false;
  const int n = dom->lay[dom->depth].n;
  if (n % GX) return false;
  for (int d = 0; d < 6; d++)
    if (dom->side[d] != GFSHIP_SIDE_PERIODIC) return false;
  return true;
}

// boxes of a lattice (PERIODIC and GfsBoundaryMpi sides only) with the library's own communicator:
// the tiled kernels with the states beyond the MPI sides exchanged in one message per side
bool godunov_fused_mpi_supported (const gfship_domain * dom)
{
  if (dom->dim != 3 || dom->no_fused_godunov || dom->no_fused_mpi || !dom->comm) return false;
  const int n = dom->lay[dom->depth].n;
  if (n % GX) return false;
  bool ext = false;
  for (int d = 0; d < 6; d++) {
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL) ext = true;
    else if (dom->side[d] != GFSHIP_SIDE_PERIODIC) return false;
  }
  return ext;
}

// send / receive buffers of the states beyond the MPI sides (3 n^2 doubles per side)
static int ghost_fv (gfship_domain * dom, GhostFv * G)
{
  if (int r = multi_buffers (dom)) return r;
  for (int d = 0; d < 6; d++) {
    G->r[d] = nullptr; G->s[d] = nullptr;
    if (dom->side[d] != GFSHIP_SIDE_EXTERNAL) continue;
    G->s[d] = dom->gfv_send[d];
    G->r[d] = dom->gfv_recv[d];
  }
  return GFSHIP_OK;
}

int launch_predict_un_fused (gfship_domain * dom, double * const u[3], double dt, int gradient,
			     const double visc[3], double * const un[3], double * div, double div_dt,
			     bool * div_done)
{
  const Layout & L = dom->lay[dom->depth];
  Visc3 vs;
  for (int c = 0; c < 3; c++) { vs.d[c] = visc[c]; vs.g[c] = dom->src[c]; }
  /* VS: the velocity components may have MAC sources (implicit viscosity, GfsSource) */
  const bool anyv = visc[0] != 0. || visc[1] != 0. || visc[2] != 0. ||
    dom->src[0] != 0. || dom->src[1] != 0. || dom->src[2] != 0.;
  if (div_done) *div_done = false;
  static const bool sweep_ok = getenv ("GFSHIP_NO_ADVECT_SWEEP") == nullptr;
  if (!dom->has_external && sweep_ok && L.n % SWX == 0 && L.n % SWY == 0 && L.n % SWZ == 0) {
    /* periodic box: the sweep along z, with the divergence of the MAC projection that follows */
    const dim3 sgrid (L.n/SWX, L.n/SWY, L.n/SWZ);
    const bool dv = div != nullptr && div_dt != 0.;
    GhostFv G0;
    for (int d = 0; d < 6; d++) { G0.r[d] = nullptr; G0.s[d] = nullptr; }
#define PS(VL_, VS_) do { \
      if (dv) hipLaunchKernelGGL ((predict_un_sweep_kernel<VL_, VS_, true, false>), sgrid, dim3 (SWN + SW_RING), 0, \
				  dom->stream, L, c3 (u), m3 (un), dt, vs, div, div_dt, G0); \
      else hipLaunchKernelGGL ((predict_un_sweep_kernel<VL_, VS_, false, false>), sgrid, dim3 (SWN + SW_RING), 0, \
			       dom->stream, L, c3 (u), m3 (un), dt, vs, div, div_dt, G0); } while (0)
    if (gradient) { if (anyv) PS (true, true); else PS (true, false); }
    else          { if (anyv) PS (false, true); else PS (false, false); }
#undef PS
    GFSHIP_HIP (hipGetLastError ());
    if (div_done) *div_done = dv;
    return GFSHIP_OK;
  }
  const dim3 grid (L.n/GX, L.n/GY, L.n/GZ);
  GhostFv G;
  for (int d = 0; d < 6; d++) { G.r[d] = nullptr; G.s[d] = nullptr; }
  const bool mpi = dom->has_external;
  if (mpi) {
    int r = ghost_fv (dom, &G);
    if (r) return r;
    const dim3 bgrid ((L.n*L.n + 255)/256, 6);
    CPtr3 none = { { nullptr, nullptr, nullptr } };
#define BK(VL_, VS_) hipLaunchKernelGGL ((boundary_face_values_kernel<true, VL_, VS_>), bgrid, dim3 (256), 0, 					 dom->stream, L, c3 (u), none, dt, vs, G)
    if (gradient) { if (anyv) BK (true, true); else BK (true, false); }
    else          { if (anyv) BK (false, true); else BK (false, false); }
#undef BK
    GFSHIP_HIP (hipGetLastError ());
    if ((r = comm_exchange_raw (dom, dom->gfv_send, dom->gfv_recv, (size_t) L.n*L.n))) return r;
    dom->n_fused_mpi++;
    static const bool mpi_sweep_ok = getenv ("GFSHIP_NO_MPI_SWEEP") == nullptr;
    if (sweep_ok && mpi_sweep_ok && L.n % SWX == 0 && L.n % SWY == 0 && L.n % SWZ == 0) {
      /* the sweep along z with the states beyond the MPI sides from the received buffers, and the divergence */
      const dim3 sgrid (L.n/SWX, L.n/SWY, L.n/SWZ);
      const bool dv = div != nullptr && div_dt != 0.;
#define PSM(VL_, VS_) do { \
	if (dv) hipLaunchKernelGGL ((predict_un_sweep_kernel<VL_, VS_, true, true>), sgrid, dim3 (SWN + SW_RING), 0, \
				    dom->stream, L, c3 (u), m3 (un), dt, vs, div, div_dt, G); \
	else hipLaunchKernelGGL ((predict_un_sweep_kernel<VL_, VS_, false, true>), sgrid, dim3 (SWN + SW_RING), 0, \
				 dom->stream, L, c3 (u), m3 (un), dt, vs, div, div_dt, G); } while (0)
      if (gradient) { if (anyv) PSM (true, true); else PSM (true, false); }
      else          { if (anyv) PSM (false, true); else PSM (false, false); }
#undef PSM
      GFSHIP_HIP (hipGetLastError ());
      if (div_done) *div_done = dv;
      return GFSHIP_OK;
    }
  }
#define PK(VL_, VS_) do { if (mpi) hipLaunchKernelGGL ((predict_un_tiled_kernel<VL_, VS_, true>), grid, dim3 (GN), 0, 							dom->stream, L, c3 (u), m3 (un), dt, vs, G);     else hipLaunchKernelGGL ((predict_un_tiled_kernel<VL_, VS_, false>), grid, dim3 (GN), 0, 			     dom->stream, L, c3 (u), m3 (un), dt, vs, G); } while (0)
  if (gradient) { if (anyv) PK (true, true); else PK (true, false); }
  else          { if (anyv) PK (false, true); else PK (false, false); }
#undef PK
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int launch_advect_fused (gfship_domain * dom, bool velocity, const double * v, double * out,
			 double * const un[3], const double * gm, const double * gc, double dt,
			 int gradient, double visc, double gsrc)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid (L.n/GX, L.n/GY, L.n/GZ);
#define AK(VE_, VL_, VS_) hipLaunchKernelGGL ((advect_tiled_kernel<VE_, VL_, VS_>), grid, dim3 (GN), 0, \
					      dom->stream, L, v, out, c3 (un), gm, gc, dt, visc, gsrc)
  const bool vs = visc != 0. || gsrc != 0.;
  if (velocity) {
    if (gradient) { if (vs) AK (true, true, true); else AK (true, true, false); }
    else          { if (vs) AK (true, false, true); else AK (true, false, false); }
  }
  else {
    if (gradient) { if (vs) AK (false, true, true); else AK (false, true, false); }
    else          { if (vs) AK (false, false, true); else AK (false, false, false); }
  }
#undef AK
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// the three velocity components at once (3-D box of periodic / MPI sides, no viscosity)
int launch_advect3_fused (gfship_domain * dom, double * const v[3], double * const out[3],
			  double * const un[3], double * const gm[3], double * const gc[3],
			  double dt, int gradient, double corr_dt, double * const uc[3])
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid (L.n/GX, L.n/GY, L.n/GZ);
  CPtr3 gcp;
  for (int c = 0; c < 3; c++) gcp.p[c] = gc ? gc[c] : nullptr;
  GhostFv G;
  for (int d = 0; d < 6; d++) { G.r[d] = nullptr; G.s[d] = nullptr; }
  Visc3 vs = { { 0., 0., 0. }, { dom->src[0], dom->src[1], dom->src[2] } };
  AdvCorr K;
  K.dt = corr_dt;
  for (int c = 0; c < 3; c++) K.uc[c] = uc ? uc[c] : nullptr;
  K.Lc = dom->lay[dom->depth > 0 ? dom->depth - 1 : 0];
  const bool corr = corr_dt != 0.;
  GFSHIP_CHECK (!corr || gc, GFSHIP_EINVAL, "the fused correction needs the centred gradient");
  const bool mpi = dom->has_external;
  if (mpi) {
    int r = ghost_fv (dom, &G);
    if (r) return r;
    const dim3 bgrid ((L.n*L.n + 255)/256, 6);
    const bool bsrc = dom->src[0] != 0. || dom->src[1] != 0. || dom->src[2] != 0.;
#define BK3(VL_, VS_) hipLaunchKernelGGL ((boundary_face_values_kernel<false, VL_, VS_>), bgrid, dim3 (256), 0, \
					  dom->stream, L, c3 (v), c3 (un), dt, vs, G)
    if (gradient) { if (bsrc) BK3 (true, true); else BK3 (true, false); }
    else          { if (bsrc) BK3 (false, true); else BK3 (false, false); }
#undef BK3
    GFSHIP_HIP (hipGetLastError ());
    if ((r = comm_exchange_raw (dom, dom->gfv_send, dom->gfv_recv, (size_t) 3*L.n*L.n))) return r;
    dom->n_fused_mpi++;
  }
  const bool srcs0 = dom->src[0] != 0. || dom->src[1] != 0. || dom->src[2] != 0.;
  static const bool sweep_ok = getenv ("GFSHIP_NO_ADVECT_SWEEP") == nullptr;
  static const bool mpi_sweep_ok = getenv ("GFSHIP_NO_MPI_SWEEP") == nullptr;
  if (mpi && sweep_ok && mpi_sweep_ok && L.n % SWX == 0 && L.n % SWY == 0 && L.n % SWZ == 0) {
    /* a box with MPI sides: the same sweep with the states beyond those sides from the received buffers */
    const dim3 sgrid (L.n/SWX, L.n/SWY, L.n/SWZ);
#define SK3M(VL_, SRC_) do { \
      if (corr) hipLaunchKernelGGL ((advect3_sweep2_kernel<VL_, SRC_, true, true>), sgrid, dim3 (SWN + SW_RING), 0, dom->stream, \
				    L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, vs, K, G); \
      else hipLaunchKernelGGL ((advect3_sweep2_kernel<VL_, SRC_, false, true>), sgrid, dim3 (SWN + SW_RING), 0, dom->stream, \
			       L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, vs, K, G); } while (0)
    if (gradient) { if (srcs0) SK3M (true, true); else SK3M (true, false); }
    else          { if (srcs0) SK3M (false, true); else SK3M (false, false); }
#undef SK3M
    GFSHIP_HIP (hipGetLastError ());
    return GFSHIP_OK;
  }
  if (!mpi && sweep_ok && L.n % SWX == 0 && L.n % SWY == 0 && L.n % SWZ == 0) {
    /* periodic box: the sweep along z (advect3_sweep_kernel) */
    const dim3 sgrid (L.n/SWX, L.n/SWY, L.n/SWZ);
    static const bool sweep1 = getenv ("GFSHIP_ADVECT_SWEEP1") != nullptr;
#define SK3(VL_, SRC_) do { \
      if (sweep1) { \
	if (corr) hipLaunchKernelGGL ((advect3_sweep_kernel<VL_, SRC_, true>), sgrid, dim3 (SWN + SW_RING), 0, dom->stream, \
				      L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, vs, K); \
	else hipLaunchKernelGGL ((advect3_sweep_kernel<VL_, SRC_, false>), sgrid, dim3 (SWN + SW_RING), 0, dom->stream, \
				 L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, vs, K); } \
      else if (corr) hipLaunchKernelGGL ((advect3_sweep2_kernel<VL_, SRC_, true, false>), sgrid, dim3 (SWN + SW_RING), 0, dom->stream, \
					 L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, vs, K, G); \
      else hipLaunchKernelGGL ((advect3_sweep2_kernel<VL_, SRC_, false, false>), sgrid, dim3 (SWN + SW_RING), 0, dom->stream, \
			       L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, vs, K, G); } while (0)
    if (gradient) { if (srcs0) SK3 (true, true); else SK3 (true, false); }
    else          { if (srcs0) SK3 (false, true); else SK3 (false, false); }
#undef SK3
    GFSHIP_HIP (hipGetLastError ());
    return GFSHIP_OK;
  }
#define AK(VL_, MPI_, SRC_) do { \
    if (corr) hipLaunchKernelGGL ((advect3_tiled_kernel<VL_, MPI_, SRC_, ADV3_WPE, true>), grid, dim3 (GN), 0, \
				  dom->stream, L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, G, vs, K); \
    else hipLaunchKernelGGL ((advect3_tiled_kernel<VL_, MPI_, SRC_, ADV3_WPE, false>), grid, dim3 (GN), 0, \
			     dom->stream, L, c3 (v), m3 (out), c3 (un), c3 (gm), gcp, dt, G, vs, K); } while (0)
  const bool srcs = dom->src[0] != 0. || dom->src[1] != 0. || dom->src[2] != 0.;
  if (srcs) {
    if (mpi) { if (gradient) AK (true, true, true); else AK (false, true, true); }
    else if (gradient) AK (true, false, true);
    else AK (false, false, true);
  }
  else if (mpi) { if (gradient) AK (true, true, false); else AK (false, true, false); }
  else if (gradient) AK (true, false, false);
  else AK (false, false, false);
#undef AK
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int launch_cfl (gfship_domain * dom, double * const u[3], double * const un[3],
		const double visc[3], double * cfl2)
{
  Visc3 vs;
  for (int c = 0; c < 3; c++) { vs.d[c] = visc ? visc[c] : 0.; vs.g[c] = dom->src[c]; }
  const Layout & L = dom->lay[dom->depth];
  long r = L.n + 1;
  long next = dom->dim == 3 ? r*r*r : r*r;
  int block = 256;
  int nblocks = (int) ((next + block - 1)/block);
  if (nblocks > 1024) nblocks = 1024;
  double * partial = dom->d_scratch;
  double * result = dom->h_pinned;      /* host memory mapped on the device: no copy kernel */
  DISPATCH (dom, cfl_partial_kernel, dim3 (nblocks), dim3 (block), L, c3 (u), c3 (un), vs, partial);
  hipLaunchKernelGGL (min_final_kernel, dim3 (1), dim3 (256), 0, dom->stream, partial, nblocks, result);
  GFSHIP_HIP (hipGetLastError ());
  GFSHIP_HIP (stream_wait_spin (dom->stream));
  *cfl2 = dom->h_pinned[0];
  return call_reduce (dom, cfl2, 1, 2);   /* gfs_all_reduce (..., MPI_MIN), src/domain.c:2921 */
}

int launch_coarse_init (gfship_domain * dom, Field * const * v, int nf)
{
  return launch_coarse_init_from (dom, v, nf, dom->depth - 1);
}

// the levels top, top - 1, ... bottom from the level above each (depth - 1 .. 0: everything)
int launch_coarse_init_from (gfship_domain * dom, Field * const * v, int nf, int top)
{
  return launch_coarse_init_levels (dom, v, nf, top, 0);
}

int launch_coarse_init_levels (gfship_domain * dom, Field * const * v, int nf, int top, int bottom)
{
  for (int f0 = 0; f0 < nf; f0 += 8) {
    int m = nf - f0 < 8 ? nf - f0 : 8;
    for (int l = top; l >= bottom; l--) {
      const Layout & Lc = dom->lay[l], & Lf = dom->lay[l + 1];
      CoarseArgs A;
      A.nf = m;
      for (int f = 0; f < m; f++) {
	A.vc[f] = v[f0 + f]->lev[l];
	A.vf[f] = v[f0 + f]->lev[l + 1];
	v[f0 + f]->zero[l] = false;
	v[f0 + f]->coarse_valid = true;
      }
      dim3 grid, block;
      cell_grid (Lc, &grid, &block);
      DISPATCH (dom, coarse_init_kernel, grid, block, Lc, Lf, A);
    }
  }
  return GFSHIP_OK;
}

int launch_velocity_divergence (gfship_domain * dom, double * const u[3], double * out)
{
  const Layout & L = dom->lay[dom->depth];
  dim3 grid, block;
  cell_grid (L, &grid, &block);
  DISPATCH (dom, velocity_divergence_kernel, grid, block, L, c3 (u), out);
  return GFSHIP_OK;
}

} // namespace gfship
