// poisson.hip -- host control of the multigrid Poisson solver: the functions behind the
// GfsPoissonSolverFunc boundary (src/poisson.h:32-38).  Control flow follows
// gfs_poisson_cycle (src/poisson.c:1109-1178) and gfs_poisson_solve (:1225-1269) statement by
// statement; every per-cell traversal is a kernel launch on the domain's stream.
#include "gfship_internal.hpp"
#include <cmath>
#include <cfloat>
#include <cstdlib>

using namespace gfship;

namespace gfship {

// weighted sweeps of this level run on the pipelined tile kernel (GFSHIP_WEIGHTED_HYPERPLANES=1: one
// launch per hyperplane, the independent implementation)
static bool weighted_pipelined (gfship_domain * dom, unsigned dimension, int level)
{
  static const bool off = getenv ("GFSHIP_WEIGHTED_HYPERPLANES") != nullptr;
  return !off && dom->relax_mode == GFSHIP_RELAX_EXACT && dimension == 3 && dom->dim == 3 &&
    skew_supported (dom, level) && !dom->force_hyperplane;
}

// gfs_relax on one level (src/poisson.c:604-632)
static int relax_level (gfship_domain * dom, unsigned dimension, int level, double omega,
			Field * u, Field * rhs, Field * dia)
{
  u->zero[level] = false;
  if (dom->weighted) {
    /* face weights from a GfsFunction alpha: the exact-order sweep with the six weights of every
       cell streamed beside u / rhs / dia by the pipelined tile kernel (3-D levels >= 32^3), in one
       launch per sweep by rows (2-D) or in LDS (small 3-D levels) */
    RelaxOp op = weighted_op (dom, level);
    if (weighted_pipelined (dom, dimension, level))
      return launch_relax_loop_skew (dom, level, u, u, rhs->lev[level], dia->lev[level], false, 1, false,
				     nullptr, nullptr, &op);
    return launch_relax_exact (dom, dimension, level, omega, u->lev[level], rhs->lev[level],
			       dia->lev[level], &op);
  }
  if (dom->relax_mode == GFSHIP_RELAX_REDBLACK)
    return launch_relax_redblack (dom, dimension, level, omega, u->lev[level], rhs->lev[level],
				  dia->lev[level]);
  if (dimension == 3 && skew_supported (dom, level) && !dom->force_hyperplane)
    return launch_relax_loop_skew (dom, level, u, u, rhs->lev[level], dia->lev[level],
				   dia->zero[level], 1, false);
  return launch_relax_exact (dom, dimension, level, omega, u->lev[level], rhs->lev[level],
			     dia->lev[level]);
}

// relax_loop, src/poisson.c:1070-1089
// correct_into: the caller's next operation is `correct' (u += dp on this level, src/poisson.c:998-
// 1003) and dp is not read again: where the level runs on the pipelined sweep the addition is done
// while the skewed copy is unpacked (*corrected set), dp's natural array is then left as it was
static int relax_loop (gfship_domain * dom, Field * dp, Field * u, unsigned dimension, int level,
		       double omega, Field * rhs, Field * dia, unsigned nrelax,
		       double * correct_into = nullptr, bool * corrected = nullptr,
		       const double * prolong_from = nullptr)
{
  int r;
  dp->zero[level] = false;
  if (dom->weighted && weighted_pipelined (dom, dimension, level) &&
      !(dom->has_external && dom->overlap && nrelax > 1)) {
    /* the whole loop in one launch on boxes without MPI sides (the weights do not change between
       the sweeps), else sweep by sweep with the BC between */
    RelaxOp op = weighted_op (dom, level);
    return launch_relax_loop_skew (dom, level, dp, u, rhs->lev[level], dia->lev[level], false, nrelax,
				   true, nullptr, nullptr, &op);
  }
  if (dom->weighted) {
    RelaxOp op = weighted_op (dom, level);
    bool done = false;
    if ((r = launch_relax_loop_small (dom, dimension, level, omega, dp, u, rhs->lev[level],
				      dia->lev[level], nrelax, &done, &op)))
      return r;
    if (done)
      return GFSHIP_OK;
    if ((r = launch_bc (dom, u, dp, level, 1))) return r;
    for (unsigned n = 0; n < nrelax - 1; n++) {
      if ((r = relax_level (dom, dimension, level, omega, dp, rhs, dia))) return r;
      if ((r = launch_bc (dom, u, dp, level, 1))) return r;
    }
    return relax_level (dom, dimension, level, omega, dp, rhs, dia);
  }
  if (dom->relax_mode == GFSHIP_RELAX_EXACT && dom->has_external && dom->overlap && nrelax > 1) {
    /* a parallel run with the domain parameter overlap = 1 (the reference's default): the first
       nrelax - 1 sweeps go through gfs_traverse_and_homogeneous_bc -- cells along the MPI sides
       first, their layers sent beside the bulk of the sweep, src/domain.c:1093-1125 -- and the
       last one is a plain gfs_domain_cell_traverse (src/poisson.c:1080-1086) */
    struct Ctx { gfship_domain * dom; Field * dp; int level; } ctx = { dom, dp, level };
    auto after_shell = [] (void * p) -> int {
      Ctx * c = (Ctx *) p;
      return bc_mpi_begin (c->dom, c->dp, c->level);
    };
    if ((r = launch_bc (dom, u, dp, level, 1))) return r;
    for (unsigned n = 0; n < nrelax - 1; n++) {
      if ((r = launch_relax_mpi_first (dom, dimension, level, omega, dp->lev[level], rhs->lev[level],
				       dia->lev[level], nullptr, after_shell, &ctx)))
	return r;
      if ((r = bc_mpi_end (dom, u, dp, level, 1))) return r;
    }
    bool done = false;
    if ((r = launch_relax_loop_small (dom, dimension, level, omega, dp, u, rhs->lev[level],
				      dia->lev[level], 1, &done)))
      return r;
    return done ? GFSHIP_OK : relax_level (dom, dimension, level, omega, dp, rhs, dia);
  }
  if (dom->relax_mode == GFSHIP_RELAX_EXACT && dimension == 3 && skew_supported (dom, level) &&
      !dom->force_hyperplane) {
    if (corrected) *corrected = correct_into != nullptr;
    return launch_relax_loop_skew (dom, level, dp, u, rhs->lev[level], dia->lev[level],
				   dia->zero[level], nrelax, true, correct_into, prolong_from);
  }
  if (dom->relax_mode == GFSHIP_RELAX_EXACT) {
    bool done = false;
    if ((r = launch_relax_loop_small (dom, dimension, level, omega, dp, u, rhs->lev[level],
				      dia->lev[level], nrelax, &done)))
      return r;
    if (done)
      return GFSHIP_OK;
  }
  /* boxes with GfsBoundaryMpi sides: the halo exchange comes between the sweeps, but a level that
     fits in LDS still runs each sweep as one launch instead of one launch per hyperplane */
  auto sweep = [&] () -> int {
    if (dom->relax_mode == GFSHIP_RELAX_EXACT && dom->has_external) {
      bool done = false;
      int rr = launch_relax_loop_small (dom, dimension, level, omega, dp, u, rhs->lev[level],
					dia->lev[level], 1, &done);
      if (rr || done) return rr;
    }
    return relax_level (dom, dimension, level, omega, dp, rhs, dia);
  };
  if ((r = launch_bc (dom, u, dp, level, 1))) return r;
  for (unsigned n = 0; n < nrelax - 1; n++) {
    if ((r = sweep ())) return r;
    if ((r = launch_bc (dom, u, dp, level, 1))) return r;
  }
  return sweep ();
}

static int norm_residual_finish (gfship_domain * dom, double dt, double s[5], gfship_norm * out);

static int norm_residual (gfship_domain * dom, double dt, Field * res, gfship_norm * out)
{
  // add_norm_residual + gfs_norm_update + dt scaling, src/domain.c:2239-2288
  const Layout & L = dom->lay[dom->depth];
  double size = 1./L.n;
  double s[5];
  int r = launch_norm (dom, dom->depth, res->lev[dom->depth], 1.*size*size, 1., s);
  if (r) return r;
  return norm_residual_finish (dom, dt, s, out);
}

// the leaf residual and its norm in one pass over the cells (rows of cells per workgroup, 1024
// workgroups; round 1's first attempt at this, one partial record per row, was slower than the two
// kernels)
static int residual_and_norm (gfship_domain * dom, double dt, Field * U, Field * R, Field * D,
			      Field * S, gfship_norm * out)
{
  const int L = dom->depth;
  S->zero[L] = false;
  if (dom->weighted) {
    int rr = launch_residual_weighted (dom, L, U->lev[L], R->lev[L], D->lev[L], S->lev[L]);
    if (rr) return rr;
    return norm_residual (dom, dt, S, out);
  }
  const double size = 1./dom->lay[L].n;
  double s[5];
  int r = launch_residual_norm (dom, L, U->lev[L], R->lev[L], D->lev[L], S->lev[L], 1.*size*size, 1., s,
				D->zero[L]);
  if (r) return r;
  return norm_residual_finish (dom, dt, s, out);
}

static int norm_residual_finish (gfship_domain * dom, double dt, double s[5], gfship_norm * out)
{
  const Layout & L = dom->lay[dom->depth];
  int r;
  double w = (double) ncells (L);
  if (dom->has_external) {
    /* domain_norm_reduce + gfs_all_reduce (bias), src/domain.c:2135-2166,2279 */
    double sums[5] = { s[0], s[1], s[2], w, s[4] }, mx[1] = { s[3] };
    if ((r = call_reduce_norm (dom, sums, 5, mx))) return r;
    s[0] = sums[0]; s[1] = sums[1]; s[2] = sums[2]; w = sums[3]; s[3] = mx[0]; s[4] = sums[4];
  }
  gfship_norm n;
  n.bias = s[0]; n.first = s[1]; n.second = s[2]; n.infty = s[3]; n.w = w;
  if (n.w > 0.) {
    n.bias /= n.w;
    n.first /= n.w;
    n.second = sqrt (n.second/n.w);
  }
  else
    n.infty = 0.;
  dt *= dt;
  n.bias = s[4]*dt;
  n.first *= dt;
  n.second *= dt;
  n.infty *= dt;
  *out = n;
  return GFSHIP_OK;
}

} // namespace gfship

extern "C" {

void gfship_multilevel_params_init (gfship_multilevel_params * par, int dim)
{
  memset (par, 0, sizeof (*par));
  par->tolerance = 1e-3;
  par->nrelax    = 4;
  par->erelax    = 1;
  par->minlevel  = 0;
  par->nitermax  = 100;
  par->nitermin  = 1;
  par->dimension = dim;
  par->weighted  = 0;
  par->beta      = 1.;
  par->omega     = 1.;
  par->function  = 0;
}

int gfship_poisson_coefficients_alpha (gfship_domain * dom, const gfship_field alpha[3])
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  if (!alpha)
    return gfship_poisson_coefficients (dom);
  double * a[3] = { nullptr, nullptr, nullptr };
  for (int c = 0; c < dom->dim; c++) {
    Field * F = get_field (dom, alpha[c]);
    if (!F) return GFSHIP_EINVAL;
    a[c] = F->lev[dom->depth];
  }
  for (int d = 0; d < 2*dom->dim; d++)
    if (dom->wf[d] < 0) {
      dom->wf[d] = gfship_field_alloc (dom, -1);
      if (dom->wf[d] < 0) return dom->wf[d];
    }
  for (int d = 0; d < 2*dom->dim; d++)
    for (int l = 0; l <= dom->depth; l++)
      dom->fields[dom->wf[d]].zero[l] = false;
  int r = launch_poisson_weights (dom, a);
  if (r) return r;
  dom->weighted = true;
  dom->weights_stamp++;         /* the skewed copies of the pipelined sweeps are stale */
  dom->unit_weights = true;     /* "coefficients have been set" for the checks of the entry points */
  return GFSHIP_OK;
}

int gfship_poisson_weights (gfship_domain * dom, int d, gfship_field * w)
{
  GFSHIP_CHECK (dom && w, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (d >= 0 && d < 2*dom->dim, GFSHIP_EINVAL, "direction %d out of range", d);
  GFSHIP_CHECK (dom->weighted, GFSHIP_EINVAL, "call gfship_poisson_coefficients_alpha first");
  *w = dom->wf[d];
  return GFSHIP_OK;
}

int gfship_poisson_coefficients (gfship_domain * dom)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  dom->weighted = false;
  // alpha = NULL on one uniform box: every leaf face weight is 1.*1.*1./1. (poisson_coeff,
  // src/poisson.c:769-797), every coarse weight the mean of four/two ones, and no cell has
  // exactly one non-boundary neighbour (face_coeff_from_below :826-853), so f[d].v == 1.
  // everywhere: the kernels carry the constant instead of six arrays per level.
  dom->unit_weights = true;
  return GFSHIP_OK;
}

int gfship_relax (gfship_domain * dom, unsigned d, int level, double omega,
		  gfship_field u, gfship_field rhs, gfship_field dia)
{
  Field * U = get_field (dom, u), * R = get_field (dom, rhs), * D = get_field (dom, dia);
  if (!U || !R || !D) return GFSHIP_EINVAL;
  GFSHIP_CHECK (level >= 0 && level <= dom->depth, GFSHIP_EINVAL, "level %d out of range", level);
  GFSHIP_CHECK (d == 2 || d == 3, GFSHIP_EINVAL, "dimension must be 2 or 3");
  GFSHIP_CHECK (dom->unit_weights, GFSHIP_EINVAL, "call gfship_poisson_coefficients first");
  for (Field * F : { U, R, D })
    if (int r = coarse_flush (dom, F, level)) return r;
  return relax_level (dom, d, level, omega, U, R, D);
}

int gfship_residual (gfship_domain * dom, unsigned d, int level,
		     gfship_field u, gfship_field rhs, gfship_field dia, gfship_field res)
{
  (void) d;
  Field * U = get_field (dom, u), * R = get_field (dom, rhs), * D = get_field (dom, dia),
    * S = get_field (dom, res);
  if (!U || !R || !D || !S) return GFSHIP_EINVAL;
  GFSHIP_CHECK (level >= 0 && level <= dom->depth, GFSHIP_EINVAL, "level %d out of range", level);
  GFSHIP_CHECK (dom->unit_weights, GFSHIP_EINVAL, "call gfship_poisson_coefficients first");
  for (Field * F : { U, R, D, S })
    if (int r = coarse_flush (dom, F, level)) return r;
  S->zero[level] = false;
  if (dom->weighted)
    return launch_residual_weighted (dom, level, U->lev[level], R->lev[level], D->lev[level],
				     S->lev[level]);
  return launch_residual (dom, level, U->lev[level], R->lev[level], D->lev[level], S->lev[level]);
}

int gfship_norm_residual (gfship_domain * dom, double dt, gfship_field res, gfship_norm * out)
{
  Field * S = get_field (dom, res);
  if (!S || !out) return GFSHIP_EINVAL;
  return norm_residual (dom, dt, S, out);
}

int gfship_norm_variable (gfship_domain * dom, gfship_field v, gfship_norm * out)
{
  // add_norm with weight = cell volume, gfs_norm_update (src/domain.c:2116-2122,2197-2232)
  Field * V = get_field (dom, v);
  if (!V || !out) return GFSHIP_EINVAL;
  const Layout & L = dom->lay[dom->depth];
  double size = 1./L.n;
  double vol = dom->dim == 3 ? size*size*size : size*size;
  double s[5];
  int r = launch_norm (dom, dom->depth, V->lev[dom->depth], 1., vol, s);
  if (r) return r;
  gfship_norm n;
  double wv = vol*(double) ncells (L);
  if (dom->has_external) {
    double sums[4] = { s[0], s[1], s[2], wv }, mx[1] = { s[3] };
    if ((r = call_reduce_norm (dom, sums, 4, mx))) return r;
    s[0] = sums[0]; s[1] = sums[1]; s[2] = sums[2]; wv = sums[3]; s[3] = mx[0];
  }
  n.bias = s[0]; n.first = s[1]; n.second = s[2]; n.infty = s[3];
  n.w = wv;
  if (n.w > 0.) {
    n.bias /= n.w;
    n.first /= n.w;
    n.second = sqrt (n.second/n.w);
  }
  else
    n.infty = 0.;
  *out = n;
  return GFSHIP_OK;
}

} // extern "C"

// gfs_poisson_cycle; with norm != nullptr the norm of the new residual (scaled by dt) comes out of
// the same pass that computes it (the solve loop asks for it right after the cycle)
static int poisson_cycle (gfship_domain * dom, gfship_multilevel_params * p,
			  gfship_field u, gfship_field rhs, gfship_field dia, gfship_field res,
			  double dt, gfship_norm * norm)
{
  GFSHIP_CHECK (dom && p, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (p->dimension == 2 || p->dimension == 3, GFSHIP_EINVAL, "dimension must be 2 or 3");
  GFSHIP_CHECK (p->nrelax > 0 && p->erelax > 0, GFSHIP_EINVAL, "nrelax and erelax must be non zero");
  GFSHIP_CHECK (dom->unit_weights, GFSHIP_EINVAL, "call gfship_poisson_coefficients first");
  /* gfs_temporary_variable: the allocation is kept between cycles (a hipMalloc/hipFree pair
     per cycle would synchronise the device); its content is fully rewritten by the cycle */
  if (dom->dp_cache < 0)
    dom->dp_cache = gfship_field_alloc (dom, -1);
  gfship_field dpf = dom->dp_cache;
  if (dpf < 0) return dpf;
  Field * U = get_field (dom, u), * R = get_field (dom, rhs), * D = get_field (dom, dia),
    * S = get_field (dom, res), * DP = get_field (dom, dpf);
  int r = GFSHIP_OK;
  bool corrected = false;      /* u += dp already done while the finest level was unpacked */
  if (!U || !R || !D || !S) return GFSHIP_EINVAL;
  const int L = dom->depth;
  unsigned minlevel = p->minlevel; /* MAX (domain->rootlevel, p->minlevel), rootlevel = 0 */
  if (minlevel > (unsigned) L) minlevel = L;
  p->depth = L;

#define TRY(x) do { if ((r = (x)) != GFSHIP_OK) goto done; } while (0)
  for (int l = 0; l <= L; l++)
    S->zero[l] = U->zero[l] = false;
  {
    /* relaxations per level (nrelax at the leaves, times erelax per level above) */
    unsigned nrl[GFSHIP_MAXLEVEL + 1];
    {
      unsigned nr = p->nrelax;
      for (int l = L; l >= 0; l--) {
	nrl[l] = nr;
	if (l > (int) minlevel) nr *= p->erelax;
      }
    }
    for (int l = 0; l <= L; l++)
      DP->zero[l] = false;
    /* the coarse end of the cycle (restrictions, relax loops and prolongations of the levels that
       fit in LDS together) in one launch where that applies */
    const int ctop = dom->weighted ? -1 : coarse_cycle_top (dom, (int) minlevel);
    /* ... or, on a box of a lattice of boxes, computed for all boxes on every rank after one
       all-gather instead of one halo exchange per sweep and level (lattice_cycle_kernel) */
    const int ltop = ctop >= 0 ? -1 : lattice_cycle_top (dom, (int) minlevel, D);
    /* compute residual on non-leafs cells (get_from_below, post-order: finest parents first) */
    for (int l = L - 1; l >= (ctop >= 0 ? ctop + 1 : ltop >= 0 ? ltop : 0); l--) {
      /* the hand-off granules of the relax loop of level l + 1, which comes on the way up, are armed
	 from here on the side stream, beside the loops of the levels below */
      if (dom->relax_mode == GFSHIP_RELAX_EXACT && p->dimension == 3 && !dom->weighted &&
	  !dom->force_hyperplane && skew_supported (dom, l + 1))
	TRY (skew_arm_ahead (dom, l + 1, nrl[l + 1]));
      /* where level l + 1 runs its relax loop on the 2 x 2 kernels the residual is copied into their
	 layout by the same pass that restricts it (and so is the restricted one, where level l does) */
      if (!dom->no_fused_restriction && !dom->skew[l + 1].rs_ready &&
	  prolongation_fused (dom, p->dimension, l + 1, nrl[l + 1]))
	TRY (launch_restrict_pack (dom, p->dimension, l + 1, S,
				   l > (int) minlevel && prolongation_fused (dom, p->dimension, l, nrl[l])));
      else
	TRY (launch_restrict (dom, p->dimension, l, S->lev[l], S->lev[l + 1]));
    }
    unsigned first = minlevel;
    if (ltop >= 0) {
      TRY (launch_lattice_cycle (dom, p->dimension, p->omega, (int) minlevel, ltop, nrl, DP, U, S));
      first = ltop + 1;
    }
    else if (ctop >= 0) {
      TRY (launch_coarse_cycle (dom, p->dimension, p->omega, (int) minlevel, ctop, nrl, DP, U, S, D));
      first = ctop + 1;
    }
    else {
      /* relax top level */
      TRY (launch_fill (dom, minlevel, DP->lev[minlevel], 0.));
      TRY (relax_loop (dom, DP, U, p->dimension, minlevel, p->omega, S, D, nrl[minlevel]));
      first = minlevel + 1;
    }
    /* relax from top to bottom */
    for (unsigned l = first; l <= p->depth; l++) {
      /* get initial guess from coarser grid -- on the levels of the 2 x 2 sweep kernels while the
	 level is copied into their layout */
      const bool fp = prolongation_fused (dom, p->dimension, (int) l, nrl[l]);
      if (!fp)
	TRY (launch_prolongate (dom, l - 1, DP->lev[l - 1], DP->lev[l]));
      TRY (relax_loop (dom, DP, U, p->dimension, l, p->omega, S, D, nrl[l],
		       (int) l == L ? U->lev[L] : nullptr, &corrected, fp ? DP->lev[l - 1] : nullptr));
    }
  }
  /* correct on leaf cells, then BC on u (gfs_traverse_and_bc ... correct, u, u) */
  if (!corrected)
    TRY (launch_correct (dom, L, U->lev[L], DP->lev[L]));
  TRY (launch_bc (dom, U, U, L, 0));
  /* compute new residual on leaf cells */
  if (norm)
    TRY (residual_and_norm (dom, dt, U, R, D, S, norm));
  else if (dom->weighted)
    TRY (launch_residual_weighted (dom, L, U->lev[L], R->lev[L], D->lev[L], S->lev[L]));
  else
    TRY (launch_residual (dom, L, U->lev[L], R->lev[L], D->lev[L], S->lev[L]));
#undef TRY
 done:
  for (int l = 0; l <= L; l++)
    dom->skew[l].rs_ready = false;
  return r;
}

extern "C" {

int gfship_poisson_cycle (gfship_domain * dom, gfship_multilevel_params * p,
			  gfship_field u, gfship_field rhs, gfship_field dia, gfship_field res)
{
  return poisson_cycle (dom, p, u, rhs, dia, res, 1., nullptr);
}

int gfship_poisson_solve (gfship_domain * dom, gfship_multilevel_params * par,
			  gfship_field lhs, gfship_field rhs, gfship_field res,
			  gfship_field dia, double dt)
{
  GFSHIP_CHECK (dom && par, GFSHIP_EINVAL, "null argument");
  Field * U = get_field (dom, lhs), * R = get_field (dom, rhs), * D = get_field (dom, dia),
    * S = get_field (dom, res);
  if (!U || !R || !D || !S) return GFSHIP_EINVAL;
  GFSHIP_CHECK (dom->unit_weights, GFSHIP_EINVAL, "call gfship_poisson_coefficients first");
  int r;
  const int L = dom->depth;
  unsigned minlevel = par->minlevel;
  par->depth = L;
  par->niter = 0;

  /* calculates the initial residual and its norm.  With nitermin >= 1 the first cycle runs whatever
     that norm is: it is then enqueued before the host waits for the norm, and both norms are read
     after the one synchronisation at the end of the cycle */
  bool deferred = par->nitermin >= 1 && par->nitermax >= 1;
  if (deferred) {
    const Layout & Ly = dom->lay[L];
    double size = 1./Ly.n;
    S->zero[L] = false;
    if (dom->weighted) {
      if ((r = launch_residual_weighted (dom, L, U->lev[L], R->lev[L], D->lev[L], S->lev[L]))) return r;
      if ((r = launch_norm_async (dom, L, S->lev[L], 1.*size*size, 1.))) return r;
    }
    else if ((r = launch_residual_norm (dom, L, U->lev[L], R->lev[L], D->lev[L], S->lev[L], 1.*size*size,
					1., nullptr, D->zero[L])))
      return r;
  }
  else {
    if ((r = residual_and_norm (dom, dt, U, R, D, S, &par->residual))) return r;
    par->residual_before = par->residual;
  }

  double res_max_before = par->residual.infty;

  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    if ((r = poisson_cycle (dom, par, lhs, rhs, dia, res, dt, &par->residual))) {
      if (deferred) {
	/* the cycle failed after the first norm was enqueued: callers that print the statistics
	   still get residual_before (and no stale residual) */
	deferred = false;
	double s0[5];
	if (hipStreamSynchronize (dom->stream) == hipSuccess) {
	  memcpy (s0, dom->h_pinned + 8, sizeof (s0));
	  if (norm_residual_finish (dom, dt, s0, &par->residual_before) == GFSHIP_OK)
	    par->residual = par->residual_before;
	}
	else
	  memset (&par->residual_before, 0, sizeof (par->residual_before)),
	    memset (&par->residual, 0, sizeof (par->residual));
      }
      break;
    }
    if (deferred) {
      /* the stream has been synchronised by the norm of the cycle: the first norm is there */
      deferred = false;
      double s0[5];
      memcpy (s0, dom->h_pinned + 8, sizeof (s0));
      if ((r = norm_residual_finish (dom, dt, s0, &par->residual_before))) break;
      res_max_before = par->residual_before.infty;
    }
    if (par->residual.infty == res_max_before) /* convergence has stopped!! */
      break;
    if (par->residual.infty > res_max_before/1.1 && par->minlevel < par->depth)
      par->minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }

  par->minlevel = minlevel;
  if (r == GFSHIP_OK)
    r = skew_check_error (dom);
  if (r == GFSHIP_OK)
    r = lattice_check_error (dom);
  return r;
}

int gfship_time_relax (gfship_domain * dom, unsigned d, int level, gfship_field u,
		       gfship_field rhs, gfship_field dia, int reps, double * ms_per_sweep)
{
  Field * U = get_field (dom, u), * R = get_field (dom, rhs), * D = get_field (dom, dia);
  if (!U || !R || !D || !ms_per_sweep || reps <= 0) return GFSHIP_EINVAL;
  GFSHIP_CHECK (level >= 0 && level <= dom->depth, GFSHIP_EINVAL, "level %d out of range", level);
  int r;
  if (dom->relax_mode == GFSHIP_RELAX_EXACT && d == 3 && skew_supported (dom, level) &&
      !dom->force_hyperplane) {
    /* the pipelined sweep kernel alone: one launch per sweep */
    U->zero[level] = false;
    r = skew_time_sweeps (dom, level, U, R->lev[level], D->lev[level], D->zero[level], reps,
			  ms_per_sweep);
    if (r == GFSHIP_OK && getenv ("GFSHIP_SKEW_STATS"))
      skew_dump_stats (dom, level);
    return r;
  }
  if ((r = relax_level (dom, d, level, 1., U, R, D))) return r; /* warm-up */
  GFSHIP_HIP (hipEventRecord (dom->ev0, dom->stream));
  for (int q = 0; q < reps; q++)
    if ((r = relax_level (dom, d, level, 1., U, R, D))) return r;
  GFSHIP_HIP (hipEventRecord (dom->ev1, dom->stream));
  GFSHIP_HIP (hipEventSynchronize (dom->ev1));
  float ms = 0.f;
  GFSHIP_HIP (hipEventElapsedTime (&ms, dom->ev0, dom->ev1));
  *ms_per_sweep = (double) ms/reps;
  if (getenv ("GFSHIP_SKEW_STATS"))
    skew_dump_stats (dom, level);
  return GFSHIP_OK;
}

int gfship_time_relax_loop (gfship_domain * dom, int level, gfship_field u, gfship_field rhs,
			    gfship_field dia, unsigned nrelax, int reps, double * ms_per_loop,
			    int * fused)
{
  return gfship_time_relax_loop_inclusive (dom, level, u, rhs, dia, nrelax, reps, ms_per_loop, fused,
					   nullptr);
}

int gfship_time_relax_loop_inclusive (gfship_domain * dom, int level, gfship_field u, gfship_field rhs,
				      gfship_field dia, unsigned nrelax, int reps,
				      double * ms_per_loop, int * fused, double * ms_inclusive)
{
  Field * U = get_field (dom, u), * R = get_field (dom, rhs), * D = get_field (dom, dia);
  if (!U || !R || !D || !ms_per_loop || !fused || reps <= 0 || nrelax == 0) return GFSHIP_EINVAL;
  GFSHIP_CHECK (level >= 0 && level <= dom->depth, GFSHIP_EINVAL, "level %d out of range", level);
  GFSHIP_CHECK (dom->relax_mode == GFSHIP_RELAX_EXACT && skew_supported (dom, level) &&
		!dom->force_hyperplane, GFSHIP_EUNSUPPORTED,
		"the pipelined sweep does not run on this level / in this mode");
  U->zero[level] = false;
  int r = skew_time_loop (dom, level, U, R->lev[level], D->lev[level], D->zero[level], nrelax,
			  reps, ms_per_loop, fused, ms_inclusive);
  if (r == GFSHIP_OK)
    r = skew_check_error (dom);
  return r;
}

} // extern "C"
