// diffusion.hip -- implicit diffusion (GfsSourceDiffusion / viscosity): kernels and host control
// behind gfs_diffusion_coefficients, gfs_diffusion_rhs, gfs_diffusion_residual,
// gfs_diffusion_cycle (src/poisson.c:1271-1690) and gfs_diffusion (src/timestep.c:735-788), for a
// constant diffusion coefficient, constant density and no solid boundaries.
//
// Under these conditions every leaf face carries the same weight w = lambda2*beta*dt*D
// (diffusion_coef, src/poisson.c:1280-1303) and every cell the same rhoc = 1.
// (diffusion_mixed_coef :1305-1348), so a level is described by one scalar instead of 2*dim
// arrays; the coarse weights are computed on the host with face_coeff_from_below's arithmetic
// (:826-853), one scalar per level.  The in-place sweeps reuse the exact-order kernels of
// poisson_kernels.hip with the diffusion cell update (RelaxOp kind 1).
#include "gfship_internal.hpp"
#include <cmath>
#include <cstdlib>

using namespace gfship;

namespace gfship {

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

// gfs_diffusion_rhs / diffusion_rhs, src/poisson.c:1392-1451:
//   f = sum_d (g.b - g.a*v) with g.a = w, g.b = w*v_nb ; rhs += (1 - beta)/beta*f/(h*h*rhoc)
template <int DIM>
__global__ void __launch_bounds__(256)
diffusion_rhs_kernel (Layout L, double w, double pbeta, const double * __restrict__ v,
		      const double * __restrict__ rhoc, double * __restrict__ rhs)
{
  CELL_LOOP_PROLOGUE (L);
  const double h = 1./L.n;
  const long off[3] = { 1, L.sy, L.sz };
  const double val = v[c];
  double f = 0.;
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    f += w*v[c + off[cc]] - w*val;
    f += w*v[c - off[cc]] - w*val;
  }
  rhs[c] += pbeta*f/(h*h*rhoc[c]);
}

// diffusion_residual, src/poisson.c:1519-1556
template <int DIM>
__global__ void __launch_bounds__(256)
diffusion_residual_kernel (Layout L, double w, const double * __restrict__ u,
			   const double * __restrict__ rhs, const double * __restrict__ rhoc,
			   double * __restrict__ res)
{
  CELL_LOOP_PROLOGUE (L);
  const double h = 1./L.n;
  const long off[3] = { 1, L.sy, L.sz };
  double a = rhoc[c];
  double ga = 0., gb = 0.;
#pragma unroll
  for (int cc = 0; cc < DIM; cc++) {
    ga += w; gb += w*u[c + off[cc]];
    ga += w; gb += w*u[c - off[cc]];
  }
  a *= h*h;
  ga = 1. + ga/a;
  gb = rhs[c] + gb/a;
  res[c] = gb - ga*u[c];
}

// gfs_get_from_below_intensive, src/fluid.c:1843-1864 (unit cell fractions): children in
// child-id order (bit0 -> +x, bit1 -> -y, bit2 -> -z), val/sa
template <int DIM>
__global__ void __launch_bounds__(256)
restrict_intensive_kernel (Layout Lc, Layout Lf, double * __restrict__ vc,
			   const double * __restrict__ vf)
{
  CELL_LOOP_PROLOGUE (Lc);
  double val = 0., sa = 0.;
#pragma unroll
  for (int id = 0; id < (1 << DIM); id++) {
    int ci = 2*i - 1 + (id & 1);
    int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
    int ck = DIM == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
    double a = 1.;
    val += vf[Lf.idx (ci, cj, ck)]*a;
    sa += a;
  }
  vc[c] = val/sa;
}

#define DLAUNCH(dom, kern, grid, block, ...) do {				\
    if ((dom)->dim == 3)						\
      hipLaunchKernelGGL (kern<3>, grid, block, 0, (dom)->stream, __VA_ARGS__); \
    else								\
      hipLaunchKernelGGL (kern<2>, grid, block, 0, (dom)->stream, __VA_ARGS__); \
    GFSHIP_HIP (hipGetLastError ());					\
  } while (0)

static RelaxOp level_op (const gfship_domain * dom, int level)
{
  RelaxOp op;
  op.kind = 1;
  op.w = dom->diff_w[level];
  double h = 1./dom->lay[level].n;
  op.h2 = h*h;
  return op;
}

// relax_loop, src/poisson.c:1070-1089, with diffusion_relax as the cell update
static int relax_loop (gfship_domain * dom, Field * dp, Field * u, int level, Field * res,
		       Field * dia, unsigned nrelax)
{
  int r;
  RelaxOp op = level_op (dom, level);
  dp->zero[level] = false;
  bool done = false;
  if ((r = launch_relax_loop_small (dom, dom->dim, level, 1., dp, u, res->lev[level],
				    dia->lev[level], nrelax, &done, &op)))
    return r;
  if (done)
    return GFSHIP_OK;
  /* 3-D levels of 32^3 and more: the pipelined tile kernels with the diffusion cell update (rhoc
     travels as their dia stream); the whole loop in one launch on boxes without MPI sides */
  static const bool pipelined = getenv ("GFSHIP_DIFFUSION_HYPERPLANES") == nullptr;
  if (pipelined && dom->dim == 3 && dom->relax_mode == GFSHIP_RELAX_EXACT && !dom->force_hyperplane &&
      skew_supported (dom, level) && !(dom->has_external && dom->overlap && nrelax > 1))
    return launch_relax_loop_skew (dom, level, dp, u, res->lev[level], dia->lev[level], false, nrelax,
				   true, nullptr, nullptr, &op);
  if ((r = launch_bc (dom, u, dp, level, 1))) return r;
  for (unsigned n = 0; n < nrelax - 1; n++) {
    if ((r = launch_relax_exact (dom, dom->dim, level, 1., dp->lev[level], res->lev[level],
				 dia->lev[level], &op)))
      return r;
    if ((r = launch_bc (dom, u, dp, level, 1))) return r;
  }
  return launch_relax_exact (dom, dom->dim, level, 1., dp->lev[level], res->lev[level],
			     dia->lev[level], &op);
}

static int residual (gfship_domain * dom, Field * U, Field * R, Field * C, Field * S)
{
  const int L = dom->depth;
  dim3 grid, block;
  cell_grid (dom->lay[L], &grid, &block);
  S->zero[L] = false;
  DLAUNCH (dom, diffusion_residual_kernel, grid, block, dom->lay[L], dom->diff_w[L],
	   U->lev[L], R->lev[L], C->lev[L], S->lev[L]);
  return GFSHIP_OK;
}

} // namespace gfship

extern "C" {

int gfship_diffusion_coefficients (gfship_domain * dom, double D, double dt, gfship_field rhoc,
				   double beta)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  GFSHIP_CHECK (beta >= 0.5 && beta <= 1., GFSHIP_EINVAL, "beta must be in [0.5,1]");
  Field * C = get_field (dom, rhoc);
  if (!C) return GFSHIP_EINVAL;
  const int L = dom->depth;
  /* diffusion_coef: v = lambda2[c]*dt*D*fraction/alpha with dt <- beta*dt */
  double cdt = beta*dt;
  dom->diff_w[L] = 1.*cdt*D*1./1.;
  /* face_coeff_from_below: mean over the FTT_CELLS/2 children on the face, in child order */
  for (int l = L - 1; l >= 0; l--) {
    double w = dom->diff_w[l + 1], sw = 0.;
    int nd = dom->dim == 3 ? 4 : 2;
    for (int m = 0; m < nd; m++)
      sw += w;
    dom->diff_w[l] = sw/nd;
  }
  /* diffusion_mixed_coef: rhoc = 1. on every cell of every level */
  for (int l = 0; l <= L; l++) {
    int r = gfship_field_fill (dom, rhoc, l, 1.*1.);
    if (r) return r;
  }
  dom->diff_ready = true;
  return GFSHIP_OK;
}

int gfship_diffusion_rhs (gfship_domain * dom, gfship_field v, gfship_field rhs,
			  gfship_field rhoc, double beta)
{
  Field * V = get_field (dom, v), * R = get_field (dom, rhs), * C = get_field (dom, rhoc);
  if (!V || !R || !C) return GFSHIP_EINVAL;
  GFSHIP_CHECK (dom->diff_ready, GFSHIP_EINVAL, "call gfship_diffusion_coefficients first");
  const int L = dom->depth;
  dim3 grid, block;
  cell_grid (dom->lay[L], &grid, &block);
  R->zero[L] = false;
  DLAUNCH (dom, diffusion_rhs_kernel, grid, block, dom->lay[L], dom->diff_w[L],
	   (1. - beta)/beta, V->lev[L], C->lev[L], R->lev[L]);
  return GFSHIP_OK;
}

int gfship_diffusion_residual (gfship_domain * dom, gfship_field u, gfship_field rhs,
			       gfship_field rhoc, gfship_field res)
{
  Field * U = get_field (dom, u), * R = get_field (dom, rhs), * C = get_field (dom, rhoc),
    * S = get_field (dom, res);
  if (!U || !R || !C || !S) return GFSHIP_EINVAL;
  GFSHIP_CHECK (dom->diff_ready, GFSHIP_EINVAL, "call gfship_diffusion_coefficients first");
  return residual (dom, U, R, C, S);
}

int gfship_diffusion_cycle (gfship_domain * dom, unsigned levelmin, unsigned depth,
			    unsigned nrelax, gfship_field u, gfship_field rhs, gfship_field rhoc,
			    gfship_field res)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  GFSHIP_CHECK (dom->diff_ready, GFSHIP_EINVAL, "call gfship_diffusion_coefficients first");
  GFSHIP_CHECK (nrelax > 0, GFSHIP_EINVAL, "nrelax must be non zero");
  GFSHIP_CHECK (depth == (unsigned) dom->depth && levelmin <= depth, GFSHIP_EINVAL,
		"levels %u..%u do not match the domain depth %d", levelmin, depth, dom->depth);
  if (dom->dp_cache < 0)
    dom->dp_cache = gfship_field_alloc (dom, -1);
  if (dom->dp_cache < 0) return dom->dp_cache;
  Field * U = get_field (dom, u), * R = get_field (dom, rhs), * C = get_field (dom, rhoc),
    * S = get_field (dom, res), * DP = get_field (dom, dom->dp_cache);
  if (!U || !R || !C || !S) return GFSHIP_EINVAL;
  const int L = dom->depth;
  int r;
#define TRY(x) do { if ((r = (x)) != GFSHIP_OK) return r; } while (0)
  /* compute residual on non-leafs cells */
  for (int l = L - 1; l >= 0; l--) {
    dim3 grid, block;
    cell_grid (dom->lay[l], &grid, &block);
    S->zero[l] = false;
    DLAUNCH (dom, restrict_intensive_kernel, grid, block, dom->lay[l], dom->lay[l + 1],
	     S->lev[l], S->lev[l + 1]);
  }
  /* relax top level */
  for (int l = 0; l <= L; l++)
    DP->zero[l] = false;
  TRY (launch_fill (dom, levelmin, DP->lev[levelmin], 0.));
  TRY (relax_loop (dom, DP, U, levelmin, S, C, 10*nrelax));
  /* relax from top to bottom */
  for (unsigned l = levelmin + 1; l <= depth; l++) {
    /* get initial guess from coarser grid */
    TRY (launch_prolongate (dom, l - 1, DP->lev[l - 1], DP->lev[l]));
    TRY (relax_loop (dom, DP, U, l, S, C, nrelax));
  }
  /* correct on leaf cells */
  U->zero[L] = false;
  TRY (launch_correct (dom, L, U->lev[L], DP->lev[L]));
  TRY (launch_bc (dom, U, U, L, 0));
  /* compute new residual on leaf cells */
  TRY (residual (dom, U, R, C, S));
#undef TRY
  return GFSHIP_OK;
}

int gfship_diffusion (gfship_domain * dom, gfship_multilevel_params * par, gfship_field v,
		      gfship_field rhs, gfship_field rhoc)
{
  GFSHIP_CHECK (dom && par, GFSHIP_EINVAL, "null argument");
  int r;
  /* res = gfs_temporary_variable (domain): kept between calls like the cycle's dp */
  if (dom->res_cache < 0)
    dom->res_cache = gfship_field_alloc (dom, -1);
  if (dom->res_cache < 0) return dom->res_cache;
  gfship_field res = dom->res_cache;
  unsigned minlevel = par->minlevel, maxlevel = dom->depth;
  if (minlevel > maxlevel) minlevel = maxlevel;
  if ((r = gfship_diffusion_residual (dom, v, rhs, rhoc, res))) return r;
  if ((r = gfship_norm_variable (dom, res, &par->residual))) return r;
  par->residual_before = par->residual;
  double res_max_before = par->residual.infty;
  par->niter = 0;
  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    if ((r = gfship_diffusion_cycle (dom, minlevel, maxlevel, par->nrelax, v, rhs, rhoc, res)))
      return r;
    if ((r = gfship_norm_variable (dom, res, &par->residual))) return r;
    if (par->residual.infty == res_max_before) /* convergence has stopped!! */
      break;
    if (par->residual.infty > res_max_before/1.1 && minlevel < maxlevel)
      minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }
  return GFSHIP_OK;
}

} // extern "C"
