/* go_diffusion.c -- oracle: implicit diffusion (SourceDiffusion / viscosity), K22 of SURVEY.md.
 * TEST INFRASTRUCTURE ONLY (see gfs_oracle.h).
 * Follows src/poisson.c:1271-1690 (gfs_diffusion_coefficients, gfs_diffusion_rhs,
 * diffusion_relax, diffusion_residual, gfs_diffusion_cycle), src/timestep.c:735-788,923-949
 * (gfs_diffusion, variable_diffusion) and src/source.c:1105-1144 (source_diffusion_value), for a
 * constant diffusion coefficient, constant density and no solids. */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "gfs_oracle.h"
#include "go_sim.h"

/* source_diffusion_value, source.c:1105-1144 (phi = v, alpha = NULL) */
double go_source_diffusion_value (GoSim * s, GoField * phi, int cell, double D)
{
  GoDomain * dom = s->dom;
  int L = dom->depth;
  const double * p = phi->lev[L];
  double ga = 0., gb = 0.;
  double v0 = p[cell];
  for (int d = 0; d < 2*dom->dim; d++) {
    /* gfs_face_gradient, fluid.c:801-805: e.a = 1., e.b = neighbour */
    double ea = 1., eb = p[cell + dom->off[L][d]];
    ga += D*ea;
    gb += D*eb;
  }
  double h = 1./dom->n[L];
  return 1.*(gb - ga*v0)/(h*h);
}

/* gfs_diffusion_coefficients, poisson.c:1350-1390: face weights w = lambda2*beta*dt*D on both
 * sides of every leaf face (diffusion_coef :1280-1303, assignment not accumulation), coarse
 * weights by face_coeff_from_below; rhoc = 1 on every cell */
static void diffusion_coefficients (GoDomain * dom, double D, double dt, double beta, GoField * rhoc)
{
  int dim = dom->dim, L = dom->depth;
  double cdt = beta*dt;
  for (int l = 0; l <= L; l++) {
    for (int d = 0; d < 2*dim; d++)
      memset (dom->w[d][l], 0, dom->size[l]*sizeof (double));
    for (size_t q = 0; q < dom->size[l]; q++)
      rhoc->lev[l][q] = 1.*1.;
  }
  size_t ncell = dim == 3 ? (size_t) dom->n[L]*dom->n[L]*dom->n[L] : (size_t) dom->n[L]*dom->n[L];
  const int * order = dom->order[L];
  for (size_t q = 0; q < ncell; q++) {
    int c = order[q];
    for (int d = 0; d < 2*dim; d += 2) {
      double v = 1.*cdt*D*1./1.;
      dom->w[d][L][c] = v;
      dom->w[d + 1][L][c + dom->off[L][d]] = v;
    }
  }
  for (int d = 1; d < 2*dim; d += 2)
    for (int q = 0; q < dom->nborder[L]; q++) {
      int c = dom->border[L][d][q];
      double v = 1.*cdt*D*1./1.;
      dom->w[d][L][c] = v;
      dom->w[d - 1][L][c + dom->off[L][d]] = v;
    }
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
	    double sw = 0.;
	    for (int m = 0; m < nd; m++) {
	      int id = dim == 3 ? cdir3[d][m] : cdir2[d][m];
	      int ci = 2*i - 1 + (id & 1);
	      int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
	      int ck = dim == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
	      sw += dom->w[d][l + 1][go_index (dom, l + 1, ci, cj, ck)];
	    }
	    sw /= nd;
	    dom->w[d][l][c] = sw;
	    int ijk[3] = { i, j, k };
	    int at_side = (d & 1) ? ijk[d/2] == 1 : ijk[d/2] == n;
	    if (sw != 0. && !at_side)
	      neighbors++;
	  }
	  if (neighbors == 1)
	    for (int d = 0; d < 2*dim; d++)
	      dom->w[d][l][c] = 0.;
	}
  }
}

/* gfs_diffusion_rhs, poisson.c:1392-1451 (uniform: gfs_face_cm_weighted_gradient gives
 * g.a = w, g.b = w*neighbour, fluid.c:1361-1366) */
static void diffusion_rhs (GoDomain * dom, GoField * v, GoField * rhs, GoField * rhoc, double beta)
{
  int L = dom->depth, n = dom->n[L];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  double pbeta = (1. - beta)/beta;
  double h = 1./n;
  for (size_t q = 0; q < ncell; q++) {
    int c = dom->order[L][q];
    double f = 0.;
    double val = v->lev[L][c];
    for (int d = 0; d < 2*dom->dim; d++) {
      double w = dom->w[d][L][c];
      double ga = 0., gb = 0.;
      if (w != 0.) { ga = w; gb = w*v->lev[L][c + dom->off[L][d]]; }
      f += gb - ga*val;
    }
    rhs->lev[L][c] += pbeta*f/(h*h*rhoc->lev[L][c]);
  }
}

/* diffusion_relax, poisson.c:1455-1484 */
static void diffusion_relax (GoDomain * dom, int level, GoField * u, GoField * res, GoField * dia)
{
  int n = dom->n[level];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  double h = 1./n;
  double * pu = u->lev[level];
  for (size_t q = 0; q < ncell; q++) {
    int c = dom->order[level][q];
    double ga = 0., gb = 0.;
    for (int d = 0; d < 2*dom->dim; d++) {
      double w = dom->w[d][level][c];
      double na = 0., nb = 0.;
      if (w != 0.) { na = w; nb = w*pu[c + dom->off[level][d]]; }
      ga += na;
      gb += nb;
    }
    double a = dia->lev[level][c]*h*h;
    ga = 1. + ga/a;
    pu[c] = (gb/a + res->lev[level][c])/ga;
  }
}

/* diffusion_residual, poisson.c:1519-1556 */
static void diffusion_residual (GoDomain * dom, GoField * u, GoField * rhs, GoField * rhoc,
				GoField * res)
{
  int L = dom->depth, n = dom->n[L];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  double h = 1./n;
  for (size_t q = 0; q < ncell; q++) {
    int c = dom->order[L][q];
    double a = rhoc->lev[L][c];
    double ga = 0., gb = 0.;
    for (int d = 0; d < 2*dom->dim; d++) {
      double w = dom->w[d][L][c];
      double na = 0., nb = 0.;
      if (w != 0.) { na = w; nb = w*u->lev[L][c + dom->off[L][d]]; }
      ga += na;
      gb += nb;
    }
    a *= h*h;
    ga = 1. + ga/a;
    gb = rhs->lev[L][c] + gb/a;
    res->lev[L][c] = gb - ga*u->lev[L][c];
  }
}

/* gfs_get_from_below_intensive on every non-leaf level (post-order), fluid.c:1843-1864 */
static void restrict_intensive (GoDomain * dom, GoField * v)
{
  int dim = dom->dim;
  for (int l = dom->depth - 1; l >= 0; l--) {
    int n = dom->n[l];
    for (int k = 1; k <= (dim == 3 ? n : 1); k++)
      for (int j = 1; j <= n; j++)
	for (int i = 1; i <= n; i++) {
	  double val = 0., sa = 0.;
	  for (int id = 0; id < (1 << dim); id++) {
	    int ci = 2*i - 1 + (id & 1);
	    int cj = 2*j - 1 + ((id & 2) ? 0 : 1);
	    int ck = dim == 3 ? 2*k - 1 + ((id & 4) ? 0 : 1) : 0;
	    double a = 1.;
	    val += v->lev[l + 1][go_index (dom, l + 1, ci, cj, ck)]*a;
	    sa += a;
	  }
	  v->lev[l][go_index (dom, l, i, j, dim == 3 ? k : 0)] = val/sa;
	}
  }
}

/* get_from_above, poisson.c:1005-1042 (same as in the Poisson cycle) */
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
	  double g1 = p[c + dom->off[level][2*cc]] - 1.*p[c];
	  double g2 = p[c + dom->off[level][2*cc + 1]] - 1.*p[c];
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

static void relax_loop (GoDomain * dom, GoField * dp, GoField * u, int level, GoField * res,
			GoField * dia, unsigned nrelax)
{
  go_homogeneous_bc (dp, u, level);
  for (unsigned n = 0; n < nrelax - 1; n++) {
    diffusion_relax (dom, level, dp, res, dia);
    go_homogeneous_bc (dp, u, level);
  }
  diffusion_relax (dom, level, dp, res, dia);
}

/* gfs_diffusion_cycle, poisson.c:1633-1690 */
static void diffusion_cycle (GoDomain * dom, unsigned levelmin, unsigned depth, unsigned nrelax,
			     GoField * u, GoField * rhs, GoField * rhoc, GoField * res)
{
  GoField * dp = go_field_new (dom, -1);
  restrict_intensive (dom, res);
  memset (dp->lev[levelmin], 0, dom->size[levelmin]*sizeof (double));
  relax_loop (dom, dp, u, levelmin, res, rhoc, 10*nrelax);
  for (unsigned l = levelmin + 1; l <= depth; l++) {
    prolongate (dom, l - 1, dp);
    relax_loop (dom, dp, u, l, res, rhoc, nrelax);
  }
  int L = dom->depth, n = dom->n[L];
  size_t ncell = dom->dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  for (size_t q = 0; q < ncell; q++) {
    int c = dom->order[L][q];
    u->lev[L][c] += dp->lev[L][c];
  }
  go_bc (u, u, L);
  diffusion_residual (dom, u, rhs, rhoc, res);
  go_field_destroy (dp);
}

/* variable_diffusion (timestep.c:923-949) + gfs_diffusion (timestep.c:735-788) */
void go_variable_diffusion (GoSim * s, GoField * v, GoField * rhs, double D, double dt,
			    GoMultilevelParams * par)
{
  GoDomain * dom = s->dom;
  GoField * rhoc = go_field_new (dom, -1);
  diffusion_coefficients (dom, D, dt, par->beta, rhoc);
  diffusion_rhs (dom, v, rhs, rhoc, par->beta);

  GoField * res = go_field_new (dom, -1);
  unsigned minlevel = par->minlevel, maxlevel = dom->depth;
  diffusion_residual (dom, v, rhs, rhoc, res);
  par->residual_before = par->residual = go_norm_variable (dom, res);
  double res_max_before = par->residual.infty;
  par->niter = 0;
  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    diffusion_cycle (dom, minlevel, maxlevel, par->nrelax, v, rhs, rhoc, res);
    par->residual = go_norm_variable (dom, res);
    if (par->residual.infty == res_max_before) /* convergence has stopped!! */
      break;
    if (par->residual.infty > res_max_before/1.1 && minlevel < maxlevel)
      minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }
  go_field_destroy (res);
  go_field_destroy (rhoc);
}
