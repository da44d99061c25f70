/* go_poisson.c -- oracle: multigrid Poisson solver of the reference, uniform grids.
 * TEST INFRASTRUCTURE ONLY (see gfs_oracle.h). */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <assert.h>
#include "gfs_oracle.h"

/* gfs_multilevel_params_init, poisson.c:70-89 */
void go_multilevel_params_init (GoMultilevelParams * par, int dim)
{
  memset (par, 0, sizeof (GoMultilevelParams));
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

/* gfs_poisson_coefficients with alpha = NULL, poisson.c:756-901.
 *   reset_coeff              :756-767  every f[d].v = 0 on every cell
 *   poisson_coeff            :769-797  v = lambda2*alpha*face_fraction/metric = 1.*1.*1./1.,
 *                                       added to both sides of every leaf face
 *   face_coeff_from_below    :826-853  non-leaf cells, post-order: mean over the children on
 *                                       side d; all six zeroed if exactly one direction has a
 *                                       non-zero weight towards a non-boundary neighbour.
 * Face traversal (ftt.c:2152-2215, ftt_internal.c): first every cell in traversal order with
 * its d = 0,2,4 faces, then the cells along sides d = 1,3,5 with their d face. */
void go_poisson_coefficients (GoDomain * dom)
{
  go_poisson_coefficients_alpha (dom, NULL);
}

/* The same with a GfsFunction alpha: alpha[c] holds gfs_function_face_value (alpha, face) of the
 * leaf faces normal to c, one array per component in the layout of a level array: the entry of
 * cell (i,j,k) is its + face along c, the ghost entry in front of the first cell its - face. */
void go_poisson_coefficients_alpha (GoDomain * dom, double * const alpha[3])
{
  int dim = dom->dim, L = dom->depth;
  for (int l = 0; l <= L; l++)
    for (int d = 0; d < 2*dim; d++)
      memset (dom->w[d][l], 0, dom->size[l]*sizeof (double));

  size_t ncell = dim == 3 ? (size_t) dom->n[L]*dom->n[L]*dom->n[L] : (size_t) dom->n[L]*dom->n[L];
  const int * order = dom->order[L];
  for (size_t q = 0; q < ncell; q++) {
    int c = order[q];
    for (int d = 0; d < 2*dim; d += 2) {
      double a = alpha ? alpha[d/2][c] : 1.;
      double v = 1.*a*1./1.;
      dom->w[d][L][c] += v;
      dom->w[d + 1][L][c + dom->off[L][d]] += v;
    }
  }
  for (int d = 1; d < 2*dim; d += 2)
    for (int q = 0; q < dom->nborder[L]; q++) {
      int c = dom->border[L][d][q];
      double a = alpha ? alpha[d/2][c + dom->off[L][d]] : 1.;
      double v = 1.*a*1./1.;
      dom->w[d][L][c] += v;
      dom->w[d - 1][L][c + dom->off[L][d]] += v;
    }

  /* children of a cell on side d, in increasing child id (ftt.h:322-341) */
  static const int cdir3[6][4] = {{1,3,5,7},{0,2,4,6},{0,1,4,5},{2,3,6,7},{0,1,2,3},{4,5,6,7}};
  static const int cdir2[4][2] = {{1,3},{0,2},{0,1},{2,3}};
  int nd = dim == 3 ? 4 : 2;
  for (int l = L - 1; l >= 0; l--) {
    int n = dom->n[l];
    for (int k = 1; k <= (dim == 3 ? n : 1); k++)
      for (int j = 1; j <= n; j++)
	for (int i = 1; i <= n; i++) {
	  size_t c = go_index (dom, l, i, j, dim == 3 ? k : 0);
	  unsigned neighbors = 0;
	  for (int d = 0; d < 2*dim; d++) {
	    double s = 0.;
	    for (int m = 0; m < nd; m++) {
	      int id = dim == 3 ? cdir3[d][m] : cdir2[d][m];
	      int ci = 2*i - 1 + (id & 1);
	      int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
	      int ck = dim == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
	      s += dom->w[d][l + 1][go_index (dom, l + 1, ci, cj, ck)];
	    }
	    s /= nd;
	    dom->w[d][l][c] = s;
	    /* neighbour exists and is not a boundary (ghost) cell */
	    int ijk[3] = { i, j, k };
	    int at_side = (d & 1) ? ijk[d/2] == 1 : ijk[d/2] == n;
	    if (s != 0. && !at_side)
	      neighbors++;
	  }
	  if (neighbors == 1)
	    for (int d = 0; d < 2*dim; d++)
	      dom->w[d][l][c] = 0.;
	}
  }
}

/* relax (poisson.c:507-530) and relax2D (:532-557) with the uniform-grid branch of
 * face_weighted_gradient (fluid.c:858-864: g->a = w; g->b = w*neighbour value),
 * applied in the reference's traversal order, in place (gfs_relax, poisson.c:604-632). */
static inline void relax_cell (const GoDomain * dom, unsigned dimension, int level, double omega,
			       double * u, const double * rhs, const double * dia, int c)
{
  double a = dia[c], b = 0.;
  int nd = 2*dom->dim;
  for (int d = 0; d < nd; d++) {
    double w = dom->w[d][level][c];
    double ga = w, gb = w*u[c + dom->off[level][d]];
    a += ga;
    b += gb;
  }
  if (dimension == 2) {
    if (a != 0.)
      u[c] = (1. - omega)*u[c] + omega*(b - rhs[c])/a;
    else
      u[c] = 0.;
  }
  else {
    if (a != 0.)
      u[c] = (b - rhs[c])/a;
    else
      u[c] = 0.;
  }
}

/* one in-place sweep of a level.  mpi_first = 0: gfs_domain_cell_traverse (traversal order).
   mpi_first = 1: the cell order of gfs_traverse_and_homogeneous_bc in a parallel run with the
   domain parameter overlap = 1, the reference's default (domain.c:682,1093-1125) */
static void relax_sweep (GoDomain * dom, unsigned d, int level, double omega,
			 GoField * u, GoField * rhs, GoField * dia, int mpi_first)
{
  int n = dom->n[level];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  double * pu = u->lev[level];
  const double * prhs = rhs->lev[level], * pdia = dia->lev[level];
  const int * order = dom->order[level];
  if (mpi_first) {
    /* parallel runs (domain.c:1093-1125): cells along GfsBoundaryMpi sides first
       (update_mpi_boundaries :1024-1049: d = 0..5, ftt_cell_traverse_boundary in traversal
       order, a cell only once -- GFS_FLAG_USED), then the remaining cells in traversal order
       (update_other_cell :1016-1022) */
    char * used = calloc (dom->size[level], 1);
    for (int s = 0; s < 2*dom->dim; s++)
      if (dom->side[s] == GO_SIDE_EXTERNAL)
	for (int q = 0; q < dom->nborder[level]; q++) {
	  int c = dom->border[level][s][q];
	  if (!used[c]) {
	    relax_cell (dom, d, level, omega, pu, prhs, pdia, c);
	    used[c] = 1;
	  }
	}
    for (size_t q = 0; q < ncell; q++)
      if (!used[order[q]])
	relax_cell (dom, d, level, omega, pu, prhs, pdia, order[q]);
    free (used);
    return;
  }
  for (size_t q = 0; q < ncell; q++)
    relax_cell (dom, d, level, omega, pu, prhs, pdia, order[q]);
}

/* gfs_relax, poisson.c:604-632: gfs_domain_cell_traverse, i.e. always the plain traversal order */
void go_relax (GoDomain * dom, unsigned d, int level, double omega,
	       GoField * u, GoField * rhs, GoField * dia)
{
  relax_sweep (dom, d, level, omega, u, rhs, dia, 0);
}

/* Same sweep but in lexicographic order x increasing, y decreasing, z decreasing.
 * Not a reference function: it exists so that tests can check the claim the device kernels
 * rest on, namely that any order compatible with the per-axis orientation of the reference's
 * traversal gives bit-identical results for a 5/7-point stencil. */
void go_relax_lexicographic (GoDomain * dom, unsigned d, int level, double omega,
			     GoField * u, GoField * rhs, GoField * dia)
{
  int n = dom->n[level];
  double * pu = u->lev[level];
  const double * prhs = rhs->lev[level], * pdia = dia->lev[level];
  for (int k = (dom->dim == 3 ? n : 1); k >= 1; k--)
    for (int j = n; j >= 1; j--)
      for (int i = 1; i <= n; i++)
	relax_cell (dom, d, level, omega, pu, prhs, pdia,
		    (int) go_index (dom, level, i, j, dom->dim == 3 ? k : 0));
}

/* residual_set / residual_set2D, poisson.c:634-678 (gfs_residual :721-747) */
void go_residual (GoDomain * dom, unsigned d, int level,
		  GoField * u, GoField * rhs, GoField * dia, GoField * res)
{
  (void) d;
  int n = dom->n[level];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const double * pu = u->lev[level], * prhs = rhs->lev[level], * pdia = dia->lev[level];
  double * pres = res->lev[level];
  const int * order = dom->order[level];
  int nd = 2*dom->dim;
  for (size_t q = 0; q < ncell; q++) {
    int c = order[q];
    double a = pdia[c], b = 0.;
    for (int dd = 0; dd < nd; dd++) {
      double w = dom->w[dd][level][c];
      double ga = w, gb = w*pu[c + dom->off[level][dd]];
      a += ga;
      b += gb;
    }
    pres[c] = prhs[c] - (b - pu[c]*a);
  }
}

/* gfs_norm_add (fluid.c:2139-2154) */
static void norm_add (GoNorm * n, double val, double weight)
{
  n->bias += weight*val;
  val = fabs (val);
  if (weight != 0. && val > n->infty)
    n->infty = val;
  n->first += weight*val;
  n->second += weight*val*val;
  n->w += weight;
}

/* gfs_norm_init (fluid.c:2107-2114), gfs_norm_update (:2162-2171) */
static void norm_init (GoNorm * n)
{
  n->bias = n->first = n->second = 0.;
  n->infty = - DBL_MAX;
  n->w = 0.;
}

static void norm_update (GoNorm * n)
{
  if (n->w > 0.0) {
    n->bias /= n->w;
    n->first /= n->w;
    n->second = sqrt (n->second/n->w);
  }
  else
    n->infty = 0.0;
}

/* gfs_domain_norm_residual on the leaves, domain.c:2239-2288 */
GoNorm go_norm_residual (GoDomain * dom, double dt, GoField * res)
{
  int L = dom->depth, n = dom->n[L];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const double * pres = res->lev[L];
  const int * order = dom->order[L];
  GoNorm nm;
  double bias = 0.;
  double size = 1./n;
  norm_init (&nm);
  for (size_t q = 0; q < ncell; q++) {
    double r = pres[order[q]];
    norm_add (&nm, r/(1.*size*size), 1.);
    bias += r;
  }
  if (dom->reduce) { /* domain_norm_reduce + gfs_all_reduce (bias), domain.c:2135-2166,2279 */
    double sums[4] = { nm.bias, nm.first, nm.second, nm.w };
    (* dom->reduce) (dom->reduce_ctx, sums, 4, 0);
    (* dom->reduce) (dom->reduce_ctx, &nm.infty, 1, 1);
    (* dom->reduce) (dom->reduce_ctx, &bias, 1, 0);
    nm.bias = sums[0]; nm.first = sums[1]; nm.second = sums[2]; nm.w = sums[3];
  }
  norm_update (&nm);
  dt *= dt;
  nm.bias = bias*dt;
  nm.first *= dt;
  nm.second *= dt;
  nm.infty *= dt;
  return nm;
}

/* gfs_domain_norm_variable on the leaves with w = NULL (domain.c:2116-2122,2197-2232):
 * volume-weighted norm */
GoNorm go_norm_variable (GoDomain * dom, GoField * v)
{
  int L = dom->depth, n = dom->n[L];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const double * pv = v->lev[L];
  const int * order = dom->order[L];
  double size = 1./n;
  double vol = dom->dim == 3 ? size*size*size : size*size;
  GoNorm nm;
  norm_init (&nm);
  for (size_t q = 0; q < ncell; q++)
    norm_add (&nm, pv[order[q]], vol);
  if (dom->reduce) {
    double sums[4] = { nm.bias, nm.first, nm.second, nm.w };
    (* dom->reduce) (dom->reduce_ctx, sums, 4, 0);
    (* dom->reduce) (dom->reduce_ctx, &nm.infty, 1, 1);
    nm.bias = sums[0]; nm.first = sums[1]; nm.second = sums[2]; nm.w = sums[3];
  }
  norm_update (&nm);
  return nm;
}

/* get_from_below_3D / _2D (poisson.c:1044-1068): parent = sum of children in child-id order,
 * halved in 3-D.  Post-order over non-leaf cells = finest parents first. */
static void restrict_residual (GoDomain * dom, unsigned dimension, GoField * v)
{
  int dim = dom->dim;
  for (int l = dom->depth - 1; l >= 0; l--) {
    int n = dom->n[l];
    double * p = v->lev[l];
    const double * f = v->lev[l + 1];
    for (int k = 1; k <= (dim == 3 ? n : 1); k++)
      for (int j = 1; j <= n; j++)
	for (int i = 1; i <= n; i++) {
	  double val = 0.;
	  for (int id = 0; id < (1 << dim); id++) {
	    int ci = 2*i - 1 + (id & 1);
	    int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
	    int ck = dim == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
	    val += f[go_index (dom, l + 1, ci, cj, ck)];
	  }
	  p[go_index (dom, l, i, j, dim == 3 ? k : 0)] = dimension == 2 ? val : val/2.;
	}
  }
}

/* get_from_above (poisson.c:1005-1042) for all cells of level `level` (parents), using
 * gfs_face_gradient at max_level = level (fluid.c:801-805: g->a = 1., g->b = neighbour).
 * ftt_cell_relative_pos: coords[n]/4. (ftt.c:327-340). */
static void prolongate (GoDomain * dom, int level, GoField * v)
{
  int dim = dom->dim, n = dom->n[level];
  const double * p = v->lev[level];
  double * f = v->lev[level + 1];
  for (int k = 1; k <= (dim == 3 ? n : 1); k++)
    for (int j = 1; j <= n; j++)
      for (int i = 1; i <= n; i++) {
	size_t c = go_index (dom, level, i, j, dim == 3 ? k : 0);
	double h[3] = { 0., 0., 0. };
	for (int cc = 0; cc < dim; cc++) {
	  double ga, gb, g1, g2;
	  ga = 1.; gb = p[c + dom->off[level][2*cc]];
	  g1 = gb - ga*p[c];
	  ga = 1.; gb = p[c + dom->off[level][2*cc + 1]];
	  g2 = gb - ga*p[c];
	  h[cc] = (g1 - g2)/2.;
	}
	for (int id = 0; id < (1 << dim); id++) {
	  int ci = 2*i - 1 + (id & 1);
	  int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
	  int ck = dim == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
	  double rel[3] = { ((id & 1) ? 1. : -1.)/4., ((id & 2) ? -1. : 1.)/4.,
			    ((id & 4) ? -1. : 1.)/4. };
	  double val = p[c];
	  for (int cc = 0; cc < dim; cc++)
	    val += rel[cc]*h[cc];
	  f[go_index (dom, level + 1, ci, cj, ck)] = val;
	}
      }
}

/* relax_loop, poisson.c:1070-1089 */
static void relax_loop (GoDomain * dom, GoField * dp, GoField * u, unsigned dimension,
			int level, double omega, GoField * rhs, GoField * dia, unsigned nrelax)
{
  go_homogeneous_bc (dp, u, level);
  /* the first nrelax - 1 sweeps go through gfs_traverse_and_homogeneous_bc (MPI-side cells
     first when the run is parallel with overlap = 1), the last one through a plain
     gfs_domain_cell_traverse (poisson.c:1080-1086) */
  for (unsigned n = 0; n < nrelax - 1; n++) {
    relax_sweep (dom, dimension, level, omega, dp, rhs, dia, dom->mpi_order);
    go_homogeneous_bc (dp, u, level);
  }
  relax_sweep (dom, dimension, level, omega, dp, rhs, dia, 0);
}

/* gfs_poisson_cycle, poisson.c:1109-1178 (rootlevel = 0 for a single box) */
void go_poisson_cycle (GoDomain * dom, GoMultilevelParams * p,
		       GoField * u, GoField * rhs, GoField * dia, GoField * res)
{
  GoField * dp = go_field_new (dom, -1);
  unsigned minlevel = p->minlevel; /* MAX (domain->rootlevel = 0, p->minlevel) */
  unsigned l, nrelax;
  int L = dom->depth;

  /* compute residual on non-leafs cells */
  restrict_residual (dom, p->dimension, res);

  /* relax top level */
  nrelax = p->nrelax;
  for (l = minlevel; l < p->depth; l++)
    nrelax *= p->erelax;

  memset (dp->lev[minlevel], 0, dom->size[minlevel]*sizeof (double)); /* gfs_cell_reset */
  relax_loop (dom, dp, u, p->dimension, minlevel, p->omega, res, dia, nrelax);
  nrelax /= p->erelax;

  /* relax from top to bottom */
  for (l = minlevel + 1; l <= p->depth; l++, nrelax /= p->erelax) {
    /* get initial guess from coarser grid */
    prolongate (dom, l - 1, dp);
    relax_loop (dom, dp, u, p->dimension, l, p->omega, res, dia, nrelax);
  }
  /* correct on leaf cells (correct, poisson.c:998-1003) then BC on u (gfs_traverse_and_bc) */
  {
    int n = dom->n[L];
    size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
    double * pu = u->lev[L];
    const double * pdp = dp->lev[L];
    for (size_t q = 0; q < ncell; q++) {
      int c = dom->order[L][q];
      pu[c] += pdp[c];
    }
  }
  go_bc (u, u, L);
  /* compute new residual on leaf cells */
  go_residual (dom, p->dimension, L, u, rhs, dia, res);

  go_field_destroy (dp);
}

/* gfs_poisson_solve, poisson.c:1225-1269 */
void go_poisson_solve (GoDomain * dom, GoMultilevelParams * par,
		       GoField * lhs, GoField * rhs, GoField * res, GoField * dia, double dt)
{
  unsigned minlevel = par->minlevel;
  par->depth = dom->depth;
  par->niter = 0;

  /* calculates the initial residual and its norm */
  go_residual (dom, par->dimension, dom->depth, lhs, rhs, dia, res);
  par->residual_before = par->residual = go_norm_residual (dom, dt, res);

  double res_max_before = par->residual.infty;

  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    go_poisson_cycle (dom, par, lhs, rhs, dia, res);
    par->residual = go_norm_residual (dom, dt, res);
    if (par->residual.infty == res_max_before) /* convergence has stopped!! */
      break;
    if (par->residual.infty > res_max_before/1.1 && par->minlevel < par->depth)
      par->minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }

  par->minlevel = minlevel;
}
