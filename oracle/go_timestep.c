/* go_timestep.c -- oracle: projection, Godunov advection and the simulation_run loop of the
 * reference on one uniform box (no solids, no sources, constant density).
 * TEST INFRASTRUCTURE ONLY (see gfs_oracle.h).
 *
 * Per-cell face state is kept literally as in GfsStateVector (fluid.h:39-52): for every leaf
 * cell (ghosts included) six normal velocities un[d] and six face values fv[d]. */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <assert.h>
#include "gfs_oracle.h"
#include "go_sim.h"

#define OPP(d) ((d) ^ 1)
#define G_MAXINT 2147483647

/* ---- face traversal, ftt.c:2152-2215 + ftt_internal.c:1-83 on a uniform box ------------
 * c < 0: FTT_XYZ: every leaf cell in traversal order with its d = 0,2,4 faces, then the cells
 * along sides d = 1,3,5 with that face.  c >= 0: direction 2c for every cell, then side 2c+1. */
typedef void (* FaceFunc) (GoSim * s, int cell, int nb, int d, void * data);

static void face_traverse (GoSim * s, int c, FaceFunc fn, void * data)
{
  GoDomain * dom = s->dom;
  int L = dom->depth, n = dom->n[L], dim = dom->dim;
  size_t ncell = dim == 3 ? (size_t) n*n*n : (size_t) n*n;
  const int * order = dom->order[L];
  if (c < 0) {
    for (size_t q = 0; q < ncell; q++)
      for (int d = 0; d < 2*dim; d += 2)
	(* fn) (s, order[q], order[q] + (int) dom->off[L][d], d, data);
    for (int d = 1; d < 2*dim; d += 2)
      for (int q = 0; q < dom->nborder[L]; q++) {
	int cell = dom->border[L][d][q];
	(* fn) (s, cell, cell + (int) dom->off[L][d], d, data);
      }
  }
  else {
    int d = 2*c;
    for (size_t q = 0; q < ncell; q++)
      (* fn) (s, order[q], order[q] + (int) dom->off[L][d], d, data);
    d = 2*c + 1;
    for (int q = 0; q < dom->nborder[L]; q++) {
      int cell = dom->border[L][d][q];
      (* fn) (s, cell, cell + (int) dom->off[L][d], d, data);
    }
  }
}

#define LEAF_LOOP(s, cellvar)						\
  for (size_t q_ = 0, nc_ = (s)->dom->dim == 3 ?			\
	 (size_t) (s)->dom->n[(s)->dom->depth]*(s)->dom->n[(s)->dom->depth]*(s)->dom->n[(s)->dom->depth] : \
	 (size_t) (s)->dom->n[(s)->dom->depth]*(s)->dom->n[(s)->dom->depth]; q_ < nc_; q_++) \
    for (int cellvar = (s)->dom->order[(s)->dom->depth][q_], once_ = 1; once_; once_ = 0)

/* ---- construction ------------------------------------------------------------------------ */

GoSim * go_sim_new (int dim, int depth, const int side[6])
{
  GoSim * s = calloc (1, sizeof (GoSim));
  s->dom = go_domain_new (dim, depth, side);
  GoDomain * dom = s->dom;
  s->p = go_field_new (dom, -1);
  s->pmac = go_field_new (dom, -1);
  for (int c = 0; c < dim; c++) {
    s->u[c] = go_field_new (dom, c);
    s->g[c] = go_field_new (dom, c);     /* gfs_variable_set_vector (g), simulation.c:455-456 */
    s->gmac[c] = go_field_new (dom, c);
  }
  size_t sz = dom->size[depth];
  for (int d = 0; d < 2*dim; d++) {
    s->un[d] = calloc (sz, sizeof (double));
    s->fv[d] = calloc (sz, sizeof (double));
  }
  go_multilevel_params_init (&s->projection_params, dim);
  go_multilevel_params_init (&s->approx_projection_params, dim);
  /* gfs_advection_params_init, advection.c:922-942 */
  s->advection_params.cfl = 0.8;
  s->advection_params.dt = 0.;
  s->advection_params.gradient = 0;
  s->advection_params.gc = 1;
  for (int c = 0; c < 3; c++) { /* diffusion_init, source.c:966-974 */
    go_multilevel_params_init (&s->diffusion_params[c], dim);
    s->diffusion_params[c].tolerance = 1e-6;
  }
  /* gfs_time_init, simulation.c:1660-1670 */
  s->t = 0.; s->end = DBL_MAX; s->i = 0; s->iend = (unsigned) G_MAXINT; s->dtmax = DBL_MAX;
  s->tnext = 0.;
  return s;
}

void go_sim_destroy (GoSim * s)
{
  if (!s) return;
  int dim = s->dom->dim;
  go_field_destroy (s->p);
  go_field_destroy (s->pmac);
  for (int c = 0; c < dim; c++) {
    go_field_destroy (s->u[c]);
    go_field_destroy (s->g[c]);
    go_field_destroy (s->gmac[c]);
  }
  for (int d = 0; d < 6; d++) {
    free (s->un[d]);
    free (s->fv[d]);
  }
  for (int t = 0; t < s->ntracers; t++)
    go_field_destroy (s->tracer[t]);
  go_domain_destroy (s->dom);
  free (s);
}

GoField * go_sim_field (GoSim * s, int which, int c)
{
  switch (which) {
  case 0: return s->p;
  case 1: return s->pmac;
  case 2: return s->u[c];
  case 3: return s->g[c];
  case 4: return s->gmac[c];
  case 5: return c < s->ntracers ? s->tracer[c] : NULL;
  }
  return NULL;
}

GoDomain * go_sim_domain (GoSim * s) { return s->dom; }
GoMultilevelParams * go_sim_projection_params (GoSim * s) { return &s->projection_params; }
GoMultilevelParams * go_sim_approx_projection_params (GoSim * s) { return &s->approx_projection_params; }
GoAdvectionParams * go_sim_advection_params (GoSim * s) { return &s->advection_params; }
double * go_sim_un (GoSim * s, int d) { return s->un[d]; }
double * go_sim_fv (GoSim * s, int d) { return s->fv[d]; }
double go_sim_time (GoSim * s) { return s->t; }
unsigned go_sim_iter (GoSim * s) { return s->i; }
void go_sim_set_time (GoSim * s, double end, double dtmax)
{
  s->end = end;
  s->dtmax = dtmax;
}

void go_sim_set_viscosity (GoSim * s, int c, double nu) { s->visc[c] = nu; }
void go_sim_set_source (GoSim * s, int c, double g) { s->src[c] = g; }
void go_sim_set_alpha (GoSim * s, GoField * const alpha[3])
{
  for (int c = 0; c < 3; c++)
    s->alpha[c] = alpha && c < s->dom->dim ? alpha[c] : NULL;
}

/* gfs_poisson_coefficients (domain, alpha, ...) of the projections, timestep.c:318,376 */
static void sim_poisson_coefficients (GoSim * s)
{
  if (s->alpha[0]) {
    double * a[3] = { NULL, NULL, NULL };
    for (int c = 0; c < s->dom->dim; c++)
      a[c] = s->alpha[c]->lev[s->dom->depth];
    go_poisson_coefficients_alpha (s->dom, a);
  }
  else
    go_poisson_coefficients (s->dom);
}

/* gfs_variable_mac_source (source.c:38-59) of a velocity component: the sum over its sources that
 * have a mac_value -- the explicit diffusion term of an implicit GfsSourceDiffusion
 * (source_diffusion_value, source.c:1105-1144) and the intensity of a GfsSource (source_value,
 * :398-403); *has = the variable has sources at all */
static double variable_mac_source (GoSim * s, GoField * v, int cell, int * has)
{
  int c = v->component;
  int present = c >= 0 && (s->visc[c] != 0. || s->src[c] != 0.);
  if (has) *has = present;
  if (!present)
    return 0.;
  double sum = 0.;
  if (s->visc[c] != 0.)
    sum += go_source_diffusion_value (s, v, cell, s->visc[c]);
  if (s->src[c] != 0.)
    sum += s->src[c];
  return sum;
}
GoMultilevelParams * go_sim_diffusion_params (GoSim * s, int c) { return &s->diffusion_params[c]; }

int go_sim_add_tracer (GoSim * s)
{
  assert (s->ntracers < GO_MAXTRACERS);
  s->tracer[s->ntracers] = go_field_new (s->dom, -1);
  s->tracer_gradient[s->ntracers] = 1;     /* gfs_center_van_leer_gradient, variable.c:429 */
  return s->ntracers++;
}

void go_sim_set_tracer_gradient (GoSim * s, int t, int gradient)
{
  assert (t >= 0 && t < s->ntracers);
  s->tracer_gradient[t] = gradient;
}

/* ---- K12: MAC velocities from centred velocities ---------------------------------------- */

/* gfs_face_reset_normal_velocity, advection.c:575-587 */
static void face_reset_un (GoSim * s, int cell, int nb, int d, void * data)
{
  s->un[OPP (d)][nb] = s->un[d][cell] = 0.;
}

/* gfs_face_interpolated_value, fluid.c:2186-2198, same-level neighbour: x1 = 1. */
static inline double face_interpolated_value (const double * v, int cell, int nb)
{
  double x1 = 1., v1 = v[nb];
  return ((x1 - 0.5)*v[cell] + 0.5*v1)/x1;
}

/* gfs_face_interpolated_normal_velocity, advection.c:549-573 */
static void face_interpolated_un (GoSim * s, int cell, int nb, int d, void * data)
{
  double u = face_interpolated_value (s->u[d/2]->lev[s->dom->depth], cell, nb);
  s->un[d][cell] = u;
  s->un[OPP (d)][nb] = u;
}

/* ---- projection --------------------------------------------------------------------------- */

typedef struct { GoField * p; GoField ** gv; double dt; } CorrectPar;

/* correct_normal_velocity, timestep.c:118-144 (face fractions = 1, FTT_FINE_FINE) */
static void correct_normal_velocity (GoSim * s, int cell, int nb, int d, void * data)
{
  CorrectPar * par = data;
  GoDomain * dom = s->dom;
  int L = dom->depth;
  const double * p = par->p->lev[L];
  double h = 1./dom->n[L];
  /* gfs_face_weighted_gradient, fluid.c:858-864 */
  double w = dom->w[d][L][cell];
  double ga = w, gb = w*p[nb];
  double dp = (gb - ga*p[cell])/h;
  if (d & 1)
    dp = - dp;
  double f = 1.;
  if (f > 0.)
    dp /= f;
  s->un[d][cell] -= dp*par->dt;
  if (par->gv)
    par->gv[d/2]->lev[L][cell] += dp*1.;
  s->un[OPP (d)][nb] -= dp*par->dt;
  if (par->gv)
    par->gv[d/2]->lev[L][nb] += dp*1.;
}

/* gfs_correct_normal_velocities, timestep.c:163-179 */
static void correct_normal_velocities (GoSim * s, GoField * p, GoField ** g, double dt)
{
  CorrectPar par = { p, g, dt };
  if (s->dom->dim == 2) { /* FTT_XY: x faces then y faces (domain.c:1763-1766) */
    face_traverse (s, 0, correct_normal_velocity, &par);
    face_traverse (s, 1, correct_normal_velocity, &par);
  }
  else
    face_traverse (s, -1, correct_normal_velocity, &par);
}

/* gfs_scale_gradients, timestep.c:60-107: both neighbours always exist here and none is a
   GfsBoundaryGradient cell */
static void scale_gradients (GoSim * s, GoField ** g)
{
  int L = s->dom->depth, dim = s->dom->dim;
  LEAF_LOOP (s, cell)
    for (int c = 0; c < dim; c++)
      g[c]->lev[L][cell] /= 2.;
  for (int c = 0; c < dim; c++)
    go_bc (g[c], g[c], L);
}

/* mac_projection, timestep.c:356-444 */
static void mac_projection (GoSim * s, GoMultilevelParams * par, double dt, GoField * p,
			    GoField ** g)
{
  GoDomain * dom = s->dom;
  int L = dom->depth, dim = dom->dim;
  double h = 1./dom->n[L];

  /* gfs_reset_gradients, timestep.c:36-58 */
  LEAF_LOOP (s, cell)
    for (int c = 0; c < dim; c++)
      g[c]->lev[L][cell] = 0.;
  /* gfs_velocity_face_sources: no sources */

  GoField * dia = go_field_new (dom, -1);
  GoField * div = go_field_new (dom, -1);
  GoField * res1 = go_field_new (dom, -1);

  sim_poisson_coefficients (s);
  /* dia = 0 on all cells: fields are born zeroed */

  /* gfs_normal_divergence, fluid.c:2310-2324 */
  LEAF_LOOP (s, cell) {
    double d_ = 0.;
    for (int d = 0; d < 2*dim; d++)
      d_ += ((d & 1) ? -1. : 1.)*s->un[d][cell]*1.;
    div->lev[L][cell] = d_*h;
  }
  /* scale_divergence, timestep.c:181-187 */
  LEAF_LOOP (s, cell)
    div->lev[L][cell] /= dt;

  go_poisson_solve (dom, par, p, div, res1, dia, dt);

  go_field_destroy (dia);
  go_field_destroy (div);
  go_field_destroy (res1);

  correct_normal_velocities (s, p, g, dt);
  scale_gradients (s, g);
}

/* gfs_correct_centered_velocities, timestep.c:498-530 */
static void correct_centered_velocities (GoSim * s, GoField ** g, double dt)
{
  int L = s->dom->depth, dim = s->dom->dim;
  LEAF_LOOP (s, cell)
    for (int c = 0; c < dim; c++)
      s->u[c]->lev[L][cell] -= g[c]->lev[L][cell]*dt;
  for (int c = 0; c < dim; c++)
    go_bc (s->u[c], s->u[c], L);
}

/* gfs_approximate_projection, timestep.c:560-596 */
void go_approximate_projection (GoSim * s, GoMultilevelParams * par, double dt, GoField * p,
				GoField ** g)
{
  face_traverse (s, -1, face_reset_un, NULL);
  face_traverse (s, -1, face_interpolated_un, NULL);
  mac_projection (s, par, dt, p, g);
  correct_centered_velocities (s, g, dt);
}

/* gfs_mac_projection, timestep.c:460-484 */
void go_mac_projection (GoSim * s, GoMultilevelParams * par, double dt, GoField * p, GoField ** g)
{
  mac_projection (s, par, dt, p, g);
}

/* ---- K13: Godunov face values -------------------------------------------------------------- */

typedef struct {
  double dt;
  GoField * v;
  int use_centered_velocity;
  int gradient;
} AdvPar;

#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#define ABS(a)    (((a) < 0) ? -(a) : (a))

/* gfs_center_gradient, fluid.c:434-475, both neighbours at the same level (x1 = x2 = 1.) */
static double center_gradient (const GoDomain * dom, int L, const double * v, int cell, int c)
{
  double v0 = v[cell];
  double x1 = 1., v1 = v[cell + dom->off[L][2*c + 1]];
  double x2 = 1., v2 = v[cell + dom->off[L][2*c]];
  return (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
}

/* generic_limiter, fluid.c:563-568 */
static double generic_limiter (double r, double beta)
{
  double v1 = MIN (r, beta), v2 = MIN (beta*r, 1.);
  v1 = MAX (0., v1);
  return MAX (v1, v2);
}

/* center_limited_gradient, fluid.c:585-614, with minmod_limiter (beta = 1.), superbee_limiter (2.),
 * sweby_limiter (1.5) (:570-583): gfs_center_minmod_gradient / _superbee_ / _sweby_ (:616-690), both
 * neighbours at the same level (x1 = x2 = 1.) */
static double center_limited_gradient (const GoDomain * dom, int L, const double * v, int cell, int c,
				       double beta)
{
  double v0 = v[cell];
  double x1 = 1., x2 = 1.;
  double v1 = v[cell + (int) dom->off[L][2*c + 1]];
  double v2 = v[cell + (int) dom->off[L][2*c]];
  double g;
  if (v0 == v1)
    g = 0.;
  else
    g = generic_limiter ((v2 - v0)*x1/((v0 - v1)*x2), beta)*(v0 - v1)/x1;
  return g;
}

/* gfs_center_van_leer_gradient, fluid.c:522-561 */
static double center_van_leer_gradient (const GoDomain * dom, int L, const double * v, int cell, int c)
{
  double x1 = 1., x2 = 1.;
  double v0 = v[cell];
  double v1 = v[cell + dom->off[L][2*c + 1]];
  double v2 = v[cell + dom->off[L][2*c]];
  double s1 = 2.*(v0 - v1);
  double s2 = 2.*(v2 - v0);
  if (s1*s2 <= 0.)
    return 0.;
  double s0 = (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
  if (ABS (s2) < ABS (s1))
    s1 = s2;
  if (ABS (s0) < ABS (s1))
    return s0;
  return s1;
}

/* transverse_term, advection.c:27-47 */
static double transverse_term (GoSim * s, const AdvPar * par, int cell, const double * msize, int c)
{
  GoDomain * dom = s->dom;
  int L = dom->depth;
  const double * v = par->v->lev[L];
  double vtan = par->use_centered_velocity ?
    s->u[c]->lev[L][cell] :
    (s->un[2*c][cell] + s->un[2*c + 1][cell])/2.;
  int fd = vtan > 0. ? 2*c + 1 : 2*c;
  /* gfs_face_gradient, fluid.c:801-805: g.a = 1., g.b = neighbour */
  double ga = 1., gb = v[cell + dom->off[L][fd]];
  double g = gb - ga*v[cell];
  if (vtan > 0.) g = - g;
  return par->dt*vtan*g/(2.*msize[c]);
}

/* gfs_cell_advected_face_values, advection.c:58-99 */
static void cell_advected_face_values (GoSim * s, const AdvPar * par, int cell)
{
  GoDomain * dom = s->dom;
  int L = dom->depth, dim = dom->dim;
  const double * v = par->v->lev[L];
  double size = 1./dom->n[L], msize[3];
  for (int c = 0; c < dim; c++)
    msize[c] = size;
  for (int c = 0; c < dim; c++) {
    double unorm = par->use_centered_velocity ?
      par->dt*s->u[c]->lev[L][cell]/msize[c] :
      par->dt*(s->un[2*c][cell] + s->un[2*c + 1][cell])/(2.*msize[c]);
    double g = par->gradient == 0 ? center_gradient (dom, L, v, cell, c) :
      par->gradient == 1 ? center_van_leer_gradient (dom, L, v, cell, c) :
      center_limited_gradient (dom, L, v, cell, c, par->gradient == 2 ? 1. : par->gradient == 3 ? 2. : 1.5);
    double vl = v[cell] + MIN ((1. - unorm)/2., 0.5)*g;
    double vr = v[cell] + MAX ((- 1. - unorm)/2., -0.5)*g;
    /* gfs_variable_mac_source (source.c:38-59): the explicit diffusion term of an implicit
       GfsSourceDiffusion (source_diffusion_value, source.c:1105-1144) */
    double msrc = variable_mac_source (s, par->v, cell, NULL);
    double src = par->dt*msrc/2.;
    double dv;
    if (dim == 2)
      dv = transverse_term (s, par, cell, msize, (c + 1) % 2);
    else {
      static const int orthogonal[3][2] = { {1, 2}, {0, 2}, {0, 1} };
      dv =  transverse_term (s, par, cell, msize, orthogonal[c][0]);
      dv += transverse_term (s, par, cell, msize, orthogonal[c][1]);
    }
    s->fv[2*c][cell]     = vl + src - dv;
    s->fv[2*c + 1][cell] = vr + src - dv;
  }
}

/* gfs_domain_face_bc (domain.c:1209-1232) for variable v on the six ghost faces:
 * symmetry boundary.c:64-74, dirichlet :275-279, neumann :354-360, periodic :1251-1258,1343-1347.
 * The ghost cell G beyond side sd holds its value in fv[OPP(sd)][G]. */
static void face_bc (GoSim * s, GoField * v)
{
  GoDomain * dom = s->dom;
  int L = dom->depth, dim = dom->dim, n = dom->n[L];
  double h = 1./n;
  int external = 0;
  for (int sd = 0; sd < 2*dim; sd++) {
    int od = OPP (sd), c = sd/2;
    if (dom->side[sd] == GO_SIDE_EXTERNAL) { external = 1; continue; }
    for (int q = 0; q < dom->nborder[L]; q++) {
      int cell = dom->border[L][sd][q];
      int G = cell + (int) dom->off[L][sd];
      if (dom->side[sd] == GO_SIDE_PERIODIC) {
	int m = cell - (int) ((n - 1)*dom->off[L][sd]); /* interior cell along the opposite side */
	s->fv[od][G] = s->fv[od][m];
	continue;
      }
      double val = 0.;
      if (v->bc[sd] != GO_BC_SYMMETRY && v->bcval[sd]) {
	/* index of the face in the n^(dim-1) array: tangential coords, first axis fastest */
	int r = n + 2;
	int ijk[3] = { cell % r, (cell/r) % r, dim == 3 ? cell/(r*r) : 1 };
	int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
	val = v->bcval[sd][(ijk[ta] - 1) + (dim == 3 ? (size_t) (ijk[tb] - 1)*n : 0)];
      }
      switch (v->bc[sd]) {
      case GO_BC_SYMMETRY:
	if (v->component == c)
	  s->fv[od][G] = s->fv[sd][cell] = 0.;
	else
	  s->fv[od][G] = s->fv[sd][cell];
	break;
      case GO_BC_DIRICHLET:
	s->fv[od][G] = s->fv[sd][cell] = val;
	break;
      case GO_BC_NEUMANN:
	s->fv[od][G] = v->lev[L][cell] + val*h/2.;
	break;
      }
    }
  }
  if (external && dom->exchange)
    for (int e = 0; e < 2*dim; e++)
      if (dom->side[e ^ 1] == GO_SIDE_EXTERNAL)
	(* dom->exchange) (dom->exchange_ctx, s->fv[e], L, 1 + e);
}

/* face_values_set, timestep.c:644-654 */
static void face_values_set (GoSim * s, const AdvPar * par)
{
  LEAF_LOOP (s, cell)
    cell_advected_face_values (s, par, cell);
  face_bc (s, par->v);
}

/* gfs_face_upwinded_value, advection.c:267-300 (FTT_FINE_FINE) */
static double face_upwinded_value (GoSim * s, int cell, int nb, int d, int centered_upwinding)
{
  double un;
  if (centered_upwinding)
    un = face_interpolated_value (s->u[d/2]->lev[s->dom->depth], cell, nb);
  else
    un = s->un[d][cell];
  if (d & 1)
    un = - un;
  return
    un > 0. ? s->fv[d][cell] :
    un < 0. ? s->fv[OPP (d)][nb] :
    (s->fv[d][cell] + s->fv[OPP (d)][nb])/2.;
}

/* gfs_face_advected_normal_velocity, advection.c:513-539 */
static void face_advected_normal_velocity (GoSim * s, int cell, int nb, int d, void * data)
{
  double u = face_upwinded_value (s, cell, nb, d, 1);
  s->un[d][cell] = u;
  s->un[OPP (d)][nb] = u;
}

/* gfs_predicted_face_velocities, timestep.c:681-717 */
void go_predicted_face_velocities (GoSim * s)
{
  int dim = s->dom->dim;
  face_traverse (s, -1, face_reset_un, NULL);
  AdvPar par = { s->advection_params.dt, NULL, 1, s->advection_params.gradient };
  for (int c = 0; c < dim; c++) {
    par.v = s->u[c];
    face_values_set (s, &par);
    face_traverse (s, c, face_advected_normal_velocity, NULL);
  }
}

/* ---- K15/K16: fluxes and update ------------------------------------------------------------ */

typedef struct { double dt; GoField * v, * fvar; GoField ** g; int c; } FluxPar;

/* gfs_face_reset, fluid.c */
static void face_reset (GoSim * s, int cell, int nb, int d, void * data)
{
  FluxPar * par = data;
  double * f = par->fvar->lev[s->dom->depth];
  f[cell] = f[nb] = 0.;
}

/* gfs_face_velocity_advection_flux, advection.c:398-435 */
static void face_velocity_advection_flux (GoSim * s, int cell, int nb, int d, void * data)
{
  FluxPar * par = data;
  int L = s->dom->depth;
  double h = 1./s->dom->n[L];
  double * f = par->fvar->lev[L];
  double flux = 1.*s->un[d][cell]*par->dt/h;
  flux *= face_upwinded_value (s, cell, nb, d, 0)
    - face_interpolated_value (par->g[par->c]->lev[L], cell, nb)*par->dt/2.;
  if (d & 1)
    flux = - flux;
  f[cell] -= flux;
  f[nb] += flux;
}

/* gfs_face_advection_flux, advection.c:356-381 */
static void face_advection_flux (GoSim * s, int cell, int nb, int d, void * data)
{
  FluxPar * par = data;
  int L = s->dom->depth;
  double h = 1./s->dom->n[L];
  double * f = par->fvar->lev[L];
  double flux = 1.*s->un[d][cell]*par->dt*face_upwinded_value (s, cell, nb, d, 0)/h;
  if (d & 1)
    flux = - flux;
  f[cell] -= flux;
  f[nb] += flux;
}

/* variable_sources, timestep.c:872-921 (Godunov, no sinking velocity, no sources) */
static void variable_sources (GoSim * s, GoField * v, GoField * sv, int gradient, int velocity_flux,
			      double dt, GoField ** gmac, GoField ** g)
{
  int L = s->dom->depth;
  GoField * fvar = go_field_new (s->dom, -1);
  FluxPar fp = { dt, v, fvar, gmac, v->component };
  AdvPar ap = { dt, v, 0, gradient };
  face_traverse (s, -1, face_reset, &fp);
  face_values_set (s, &ap);
  face_traverse (s, -1, velocity_flux ? face_velocity_advection_flux : face_advection_flux, &fp);
  /* gfs_advection_update (non-merged) into sv (par->v = sv), advection.c:817-818 */
  LEAF_LOOP (s, cell)
    sv->lev[L][cell] += fvar->lev[L][cell]/1.;
  go_field_destroy (fvar);
  if (g)
    /* add_pressure_gradient on sv, timestep.c:809-812 */
    LEAF_LOOP (s, cell)
      sv->lev[L][cell] -= g[v->component]->lev[L][cell]*dt;
  /* gfs_domain_variable_centered_sources (source.c:62-108): the sources with a centered_value -- a
     GfsSource, not the implicit diffusion */
  if (v->component >= 0 && s->src[v->component] != 0.)
    LEAF_LOOP (s, cell) {
      double sum = 0;
      sum += s->src[v->component];
      sv->lev[L][cell] += dt*sum;
    }
}

/* gfs_centered_velocity_advection_diffusion, timestep.c:976-1016 (no diffusion source) */
void go_centered_velocity_advection (GoSim * s, GoField ** gmac, GoField ** g)
{
  int dim = s->dom->dim, L = s->dom->depth;
  for (int c = 0; c < dim; c++) {
    if (s->visc[c] != 0.) {
      /* source_diffusion (v[c]): rhs = copy of v, sources into rhs, then the implicit solve
	 (timestep.c:996-1007) */
      GoField * rhs = go_field_new (s->dom, -1);
      LEAF_LOOP (s, cell)
	rhs->lev[L][cell] = s->u[c]->lev[L][cell];
      variable_sources (s, s->u[c], rhs, s->advection_params.gradient, 1, s->advection_params.dt,
			gmac, g);
      go_variable_diffusion (s, s->u[c], rhs, s->visc[c], s->advection_params.dt,
			     &s->diffusion_params[c]);
      go_field_destroy (rhs);
    }
    else
      variable_sources (s, s->u[c], s->u[c], s->advection_params.gradient, 1,
			s->advection_params.dt, gmac, g);
  }
  for (int c = 0; c < dim; c++)
    go_bc (s->u[c], s->u[c], L);
}

/* gfs_tracer_advection_diffusion, timestep.c:1028-1055 (no diffusion);
   tracers default to the van Leer gradient and gfs_face_advection_flux (variable.c:427-431) */
void go_tracer_advection (GoSim * s, GoField * t, double dt)
{
  int gradient = 1;
  for (int q = 0; q < s->ntracers; q++)
    if (s->tracer[q] == t) gradient = s->tracer_gradient[q];
  variable_sources (s, t, t, gradient, 0, dt, NULL, NULL);
  go_bc (t, t, s->dom->depth);
}

/* gfs_advance_tracers, simulation.c:405-430 */
static void advance_tracers (GoSim * s, double dt)
{
  for (int t = 0; t < s->ntracers; t++)
    go_tracer_advection (s, s->tracer[t], dt);
}

/* ---- K19: CFL and time step ---------------------------------------------------------------- */

static void minimum_mac_cfl (GoSim * s, int cell, int nb, int d, void * data)
{
  double * cfl = data;
  double un = s->un[d][cell];
  double length = 1./s->dom->n[s->dom->depth];
  if (un != 0.) {
    double cflu = length/fabs (un);
    if (cflu*cflu < *cfl)
      *cfl = cflu*cflu;
  }
}

/* gfs_domain_cfl, domain.c:2824-2923 */
double go_domain_cfl (GoSim * s)
{
  int L = s->dom->depth, dim = s->dom->dim;
  double cfl = DBL_MAX;
  double length = 1./s->dom->n[L];
  face_traverse (s, -1, minimum_mac_cfl, &cfl);
  LEAF_LOOP (s, cell)
    for (int c = 0; c < dim; c++) {
      double fm = 1.;
      double u = s->u[c]->lev[L][cell];
      if (u != 0.) {
	double cflu = length/fabs (fm*u);
	if (cflu*cflu < cfl)
	  cfl = cflu*cflu;
      }
      int has;
      double g = variable_mac_source (s, s->u[c], cell, &has);
      if (has) { /* p->v[c]->sources: acceleration scale, domain.c:2893-2901 */
	if (g != 0.) {
	  double cflg = 2.*length/fabs (fm*g);
	  if (cflg < cfl)
	    cfl = cflg;
	}
      }
    }
  if (s->dom->reduce) /* gfs_all_reduce (domain, p.cfl, MPI_DOUBLE, MPI_MIN), domain.c:2921 */
    (* s->dom->reduce) (s->dom->reduce_ctx, &cfl, 1, 2);
  return sqrt (cfl);
}

/* gfs_simulation_set_timestep, simulation.c:1569-1633; the only event time is `end` */
void go_set_timestep (GoSim * s)
{
  double t = s->t;
  double cfl = s->advection_params.cfl; /* min_cfl: no tracer has a smaller cfl here */
  if (cfl < DBL_MAX)
    s->advection_params.dt = cfl*go_domain_cfl (s);
  else
    s->advection_params.dt = G_MAXINT;
  if (s->advection_params.dt > s->dtmax)
    s->advection_params.dt = s->dtmax;

  double tnext = G_MAXINT;
  if (s->end < tnext)
    tnext = s->end;

  double n = ceil ((tnext - t)/s->advection_params.dt);
  if (n > 0. && n < G_MAXINT) {
    s->advection_params.dt = (tnext - t)/n;
    if (n == 1.)
      s->tnext = tnext;
    else
      s->tnext = t + s->advection_params.dt;
  }
  else
    s->tnext = t + s->advection_params.dt;

  if (s->advection_params.dt < 1e-9)
    s->advection_params.dt = 1e-9;
}

/* ---- K18: gfs_cell_coarse_init with gfs_get_from_below_intensive -------------------------- */

static void coarse_init_field (GoDomain * dom, GoField * v)
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

/* adaptive.c:43-58: every variable of the domain */
void go_coarse_init (GoSim * s)
{
  int dim = s->dom->dim;
  coarse_init_field (s->dom, s->p);
  coarse_init_field (s->dom, s->pmac);
  for (int c = 0; c < dim; c++)
    coarse_init_field (s->dom, s->u[c]);
  for (int t = 0; t < s->ntracers; t++)
    coarse_init_field (s->dom, s->tracer[t]);
}

/* ---- simulation_run, simulation.c:432-557 --------------------------------------------------- */

/* everything before the loop: gfs_simulation_init (BCs on all variables), first time step,
   initial approximate projection (simulation.c:458-476) */
void go_sim_start (GoSim * s)
{
  int L = s->dom->depth, dim = s->dom->dim;
  go_bc (s->p, s->p, L);
  go_bc (s->pmac, s->pmac, L);
  for (int c = 0; c < dim; c++)
    go_bc (s->u[c], s->u[c], L);
  for (int t = 0; t < s->ntracers; t++)
    go_bc (s->tracer[t], s->tracer[t], L);
  go_coarse_init (s);

  go_set_timestep (s);
  if (s->i == 0) {
    go_approximate_projection (s, &s->approx_projection_params, s->advection_params.dt,
			       s->p, s->g);
    go_set_timestep (s);
    advance_tracers (s, s->advection_params.dt/2.);
  }
  else {
    /* gfs_update_gradients, timestep.c:306-322 (simulation.c:474-475: a simulation read back
       from a snapshot, time.i > 0) */
    GoField ** g = s->g;
    LEAF_LOOP (s, cell)                         /* gfs_reset_gradients */
      for (int c = 0; c < dim; c++)
	g[c]->lev[L][cell] = 0.;
    sim_poisson_coefficients (s);               /* gfs_poisson_coefficients (alpha) */
    correct_normal_velocities (s, s->p, g, 0.); /* dt = 0.: un -= dp*0., g += dp */
    scale_gradients (s, g);
  }
}

/* GfsTime { i = .. t = .. } of a snapshot, gfs_time_read simulation.c:1687-1725 */
void go_sim_restart (GoSim * s, double t, unsigned i)
{
  s->t = s->tnext = t;
  s->i = i;
}

/* one iteration of the while loop, simulation.c:479-548 (events are the caller's business) */
void go_sim_step (GoSim * s)
{
  GoField ** gc = s->advection_params.gc ? s->g : NULL;
  assert (gc); /* gc = 0 (g aliased to gmac) is not restated */

  go_predicted_face_velocities (s);

  /* gfs_variables_swap (p, pmac) around the MAC projection (simulation.c:498-503,
     variable.c:234-243: only the storage index is swapped, so the projection runs on Pmac's
     data with the boundary conditions of P) */
  {
    GoField alias = *s->p;
    memcpy (alias.lev, s->pmac->lev, sizeof (alias.lev));
    go_mac_projection (s, &s->projection_params, s->advection_params.dt/2., &alias, s->gmac);
  }

  go_centered_velocity_advection (s, s->gmac, s->i > 0 ? gc : s->gmac);
  correct_centered_velocities (s, s->i > 0 ? gc : s->gmac, - s->advection_params.dt);

  go_coarse_init (s);

  go_approximate_projection (s, &s->approx_projection_params, s->advection_params.dt,
			     s->p, s->g);

  s->t = s->tnext;
  s->i++;

  go_set_timestep (s);
  advance_tracers (s, s->advection_params.dt);
}

/* loop body of advection_run (simulation.c:2078-2111) for a GfsAdvection simulation whose velocity
 * comes from a GfsVariableStreamFunction (the MAC velocities un are kept as they are): coarse
 * values, time step, tracers */
void go_advection_step (GoSim * s)
{
  go_coarse_init (s);
  go_set_timestep (s);
  advance_tracers (s, s->advection_params.dt);
  s->t = s->tnext;
  s->i++;
}

/* gfs_divergence (fluid.c:2357-2376) on every leaf: derived variable "Divergence" */
void go_divergence (GoSim * s, GoField * out)
{
  GoDomain * dom = s->dom;
  int L = dom->depth, dim = dom->dim;
  double h = 1./dom->n[L];
  LEAF_LOOP (s, cell) {
    double div = 0.;
    for (int d = 0; d < 2*dim; d++)
      div += 1.*((d & 1) ? -1. : 1.)*
	face_interpolated_value (s->u[d/2]->lev[L], cell, cell + (int) dom->off[L][d]);
    out->lev[L][cell] = div/(1.*h);
  }
}
