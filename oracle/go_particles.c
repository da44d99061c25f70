/* go_particles.c -- oracle: Lagrangian tracers (GfsParticle in a GfsParticleList).
 * TEST INFRASTRUCTURE ONLY (see gfs_oracle.h).
 *
 * Follows src/particle.c:31-44 (gfs_particle_event), src/domain.c:2623-2638,2764-2788
 * (gfs_domain_locate, gfs_domain_advect_point), src/ftt.c:1535-1574 (ftt_cell_locate),
 * src/fluid.c:2596-2710,2938-3101 (corner interpolation) and
 * modules/particulatecommon.c:955-1015,3058-3395 (list event, ray march to the exit face,
 * periodic wrap).  The reference holds no test or golden data for any of this: parity of this
 * file is UNPINNED by the reference; it is pinned only by analytic properties (tests/). */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "gfs_oracle.h"
#include "go_sim.h"
#include "go_particles.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846   /* math.h, as used by the reference */
#endif

/* ftt_cell_locate on the box root (unit box centred on the origin), max_depth = -1:
 * returns 0 if outside, else fills the 1-based cell coordinates of the leaf */
int go_locate (const GoDomain * dom, const double target[3], int ijk[3])
{
  int dim = dom->dim, L = dom->depth;
  double pos[3] = { 0., 0., 0. };
  double size = 1./2.;
  for (int c = 0; c < dim; c++)
    if (target[c] > pos[c] + size || target[c] < pos[c] - size)
      return 0;
  int q[3] = { 0, 0, 0 }; /* 0-based coordinates while descending */
  for (int l = 0; l < L; l++) {
    /* index[z > pz][y > py][x > px] = {{{6,7},{4,5}},{{2,3},{0,1}}}: child id bit0 = x > px,
       bit1 = !(y > py), bit2 = !(z > pz); coords[n] = (+-1, -+1, -+1) (ftt.c:301-316) */
    size /= 2.;
    for (int c = 0; c < dim; c++) {
      int up = target[c] > pos[c];
      q[c] = 2*q[c] + up;
      pos[c] += (up ? 1. : -1.)*size;
    }
  }
  for (int c = 0; c < 3; c++)
    ijk[c] = c < dim ? q[c] + 1 : 0;
  return 1;
}

/* does the cell (ci,cj,ck) (coordinates in 0..n+1) exist?  Interior cells and the one-cell ghost
 * layer of each side exist; edge and corner ghosts do not (the boundary trees of the reference are
 * flattened against one box side, boundary.c:576-685) */
static int cell_exists (int n, int dim, const int c[3], int * is_boundary)
{
  int out = 0;
  for (int a = 0; a < dim; a++)
    if (c[a] < 1 || c[a] > n) {
      if (c[a] < 0 || c[a] > n + 1) return 0;
      out++;
    }
  *is_boundary = out > 0;
  return out <= 1;
}

/* gfs_cell_corner_value (fluid.c:3081-3101) with gfs_cell_corner_interpolator (:2983-3068) on a
 * uniform grid: the cells sharing the corner are n[m] = cell + (m&1) d0 + (m&2) d1 + (m&4) d2
 * (path table :2938-2954), weights 1/(distance + 1e-12) normalised, domain corners lose the
 * central cell. sgn[c] = +1/-1: direction of the corner along axis c. */
static double corner_value (const GoDomain * dom, const double * v, const int cell[3],
			    const int sgn[3])
{
  int dim = dom->dim, L = dom->depth, n = dom->n[L];
  int ncells = 1 << dim;
  double h = 1./n;
  double w[8], val[8];
  int cnt = 0, boundaries = 0;
  double wsum = 0.;
  for (int m = 0; m < ncells; m++) {
    int c[3] = { cell[0], cell[1], cell[2] };
    for (int a = 0; a < dim; a++)
      if (m & (1 << a)) c[a] += sgn[a];
    int isb;
    if (!cell_exists (n, dim, c, &isb))
      continue;
    double dist = h*(dim == 2 ? 0.707106781185 : 0.866025403785);
    double a_ = 1./(dist + 1e-12);
    w[cnt] = a_;
    val[cnt] = v[go_index (dom, L, c[0], c[1], dim == 3 ? c[2] : 0)];
    cnt++;
    wsum += a_;
    if (isb) boundaries++;
  }
  int first = 0;
  if (cnt == dim + 1 && boundaries == dim) {
    /* remove central cell from interpolator */
    wsum -= w[0];
    first = 1;
  }
  double scale = 1./wsum;
  double r = 0.;
  for (int m = first; m < cnt; m++) {
    double wm = w[m]*scale;
    r += wm*val[m];
  }
  return r;
}

/* gfs_interpolate (fluid.c:2697-2710) = gfs_cell_corner_values (:2617-2631) +
 * gfs_interpolate_from_corners (:2640-2683) */
double go_interpolate (const GoDomain * dom, const double * v, const int cell[3], const double p_[3])
{
  int dim = dom->dim, L = dom->depth;
  double f[9];
  /* corner[] tables, fluid.c:2588-2605 */
  static const int c2[4][3] = { {-1,-1,0}, {1,-1,0}, {1,1,0}, {-1,1,0} };
  static const int c3[8][3] = { {-1,-1,1}, {1,-1,1}, {1,1,1}, {-1,1,1},
				{-1,-1,-1}, {1,-1,-1}, {1,1,-1}, {-1,1,-1} };
  int nc = 4*(dim - 1);
  for (int i = 0; i < nc; i++)
    f[i] = corner_value (dom, v, cell, dim == 2 ? c2[i] : c3[i]);
  f[nc] = v[go_index (dom, L, cell[0], cell[1], dim == 3 ? cell[2] : 0)];

  double o[3];
  go_cell_pos (dom, L, cell[0], cell[1], cell[2], o);
  double size = (1./dom->n[L])/2.;
  double p[3];
  p[0] = (p_[0] - o[0])/size;
  p[1] = (p_[1] - o[1])/size;
  if (dim == 2) {
    double x = (p[0] + p[1])/2., y = (p[1] - p[0])/2., val = f[4];
    if (x > 0.)
      val += x*(f[2] - f[4]);
    else
      val -= x*(f[0] - f[4]);
    if (y > 0.)
      val += y*(f[3] - f[4]);
    else
      val -= y*(f[1] - f[4]);
    return val;
  }
  double c[8];
  p[2] = (p_[2] - o[2])/size;
  c[0] = - f[0] + f[1] + f[2] - f[3] - f[4] + f[5] + f[6] - f[7];
  c[1] = - f[0] - f[1] + f[2] + f[3] - f[4] - f[5] + f[6] + f[7];
  c[2] =   f[0] + f[1] + f[2] + f[3] - f[4] - f[5] - f[6] - f[7];
  c[3] =   f[0] - f[1] + f[2] - f[3] + f[4] - f[5] + f[6] - f[7];
  c[4] = - f[0] + f[1] + f[2] - f[3] + f[4] - f[5] - f[6] + f[7];
  c[5] = - f[0] - f[1] + f[2] + f[3] + f[4] + f[5] - f[6] - f[7];
  c[6] =   f[0] - f[1] + f[2] - f[3] - f[4] + f[5] - f[6] + f[7];
  c[7] =   f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7];
  return (c[0]*p[0] + c[1]*p[1] + c[2]*p[2] +
	  c[3]*p[0]*p[1] + c[4]*p[0]*p[2] + c[5]*p[1]*p[2] +
	  c[6]*p[0]*p[1]*p[2] +
	  c[7])/8.;
}

/* gfs_domain_advect_point, domain.c:2764-2788 */
void go_advect_point (GoSim * s, double p[3], double dt)
{
  GoDomain * dom = s->dom;
  int dim = dom->dim, L = dom->depth;
  double p0[3] = { p[0], p[1], p[2] }, p1[3] = { p[0], p[1], p[2] };
  int cell[3];
  if (!go_locate (dom, p0, cell))
    return;
  for (int c = 0; c < dim; c++)
    p1[c] += dt*go_interpolate (dom, s->u[c]->lev[L], cell, p0)/2.;
  if (!go_locate (dom, p1, cell))
    return;
  for (int c = 0; c < dim; c++)
    p[c] += dt*go_interpolate (dom, s->u[c]->lev[L], cell, p1);
}

/* check_intersetion, particulatecommon.c:3058-3146: first direction d whose face of the cell
 * (centre cellpos, size) is crossed by the segment p0 -> p1 */
static int check_intersection (int dim, const double cellpos[3], const double p0[3],
			       const double p1[3], int * dstore, double size)
{
  for (int d = 0; d < 2*dim; d++) {
    double normal = (double) (d ^ 1) - (double) d;
    int c = d/2;
    if ((p1[c] - p0[c]) != 0 && normal*(p1[c] - p0[c]) > 0) {
      double t = (cellpos[c] + normal*size*0.5 - p0[c])/(p1[c] - p0[c]);
      int inside = 1;
      for (int a = 0; a < dim; a++)
	if (a != c) {
	  double pa = p0[a] + t*(p1[a] - p0[a]);
	  if (!((pa - cellpos[a] + size*0.5)*(pa - cellpos[a] - size*0.5) <= 0))
	    inside = 0;
	}
      if (inside && t*(t - 1) <= 0) {
	*dstore = d;
	return 1;
      }
    }
  }
  return 0; /* "Intersection finding algorithm failed": dstore is left as it was */
}

/* boundarycell, particulatecommon.c:3149-3186: march from the cell of pos_old along the path
 * until the next cell would be a boundary (ghost) cell; returns the exit direction */
static int boundary_direction (const GoDomain * dom, const double pos_old[3], const double pos[3])
{
  int dim = dom->dim, L = dom->depth, n = dom->n[L];
  double size = 1./n;
  int cell[3];
  int ok = go_locate (dom, pos_old, cell);
  assert (ok);
  int d = 0; /* uninitialised in the reference when the search fails; 0 here */
  for (int guard = 0; guard < 4*n; guard++) {
    double cellpos[3];
    go_cell_pos (dom, L, cell[0], cell[1], cell[2], cellpos);
    check_intersection (dim, cellpos, pos_old, pos, &d, size);
    int c = d/2, step = (d & 1) ? -1 : 1;
    int next = cell[c] + step;
    if (next < 1 || next > n)
      return d; /* neighbour is a boundary cell */
    cell[c] = next;
  }
  return d;
}

GoParticles * go_particles_new (int np, const double * pos, const unsigned * id)
{
  GoParticles * pl = calloc (1, sizeof (GoParticles));
  pl->n = pl->cap = np;
  pl->pos = malloc (3*(size_t) np*sizeof (double));
  pl->pos_old = malloc (3*(size_t) np*sizeof (double));
  pl->id = malloc ((size_t) np*sizeof (unsigned));
  memcpy (pl->pos, pos, 3*(size_t) np*sizeof (double));
  memcpy (pl->pos_old, pos, 3*(size_t) np*sizeof (double));
  memcpy (pl->id, id, (size_t) np*sizeof (unsigned));
  return pl;
}

void go_particles_destroy (GoParticles * pl)
{
  if (!pl) return;
  free (pl->pos); free (pl->pos_old); free (pl->id);
  free (pl->vel); free (pl->force); free (pl->mass); free (pl->volume);
  for (int c = 0; c < 3; c++) if (pl->uold[c]) go_field_destroy (pl->uold[c]);
  for (int d = 0; d < 6; d++) free (pl->out[d]);
  free (pl);
}

/* doubles per migrating particle: pos, pos_old, id (+ vel, mass, volume, force for particulates) */
int go_particles_record_size (const GoParticles * pl) { return pl->particulate ? 15 : 7; }

/* the packet of particles sent through side d (go_particles_record_size doubles per particle) */
int go_particles_outbox (GoParticles * pl, int d, double ** rec)
{
  *rec = pl->out[d];
  return pl->nout[d];
}

void go_particles_clear_outbox (GoParticles * pl)
{
  for (int d = 0; d < 6; d++) pl->nout[d] = 0;
}

/* mpi_rcv_particle, particulatecommon.c:3224-3245: the received particles join the list */
void go_particles_append (GoParticles * pl, int n, const double * rec)
{
  int rs = go_particles_record_size (pl);
  if (pl->n + n > pl->cap) {
    pl->cap = 2*(pl->n + n);
    pl->pos = realloc (pl->pos, 3*(size_t) pl->cap*sizeof (double));
    pl->pos_old = realloc (pl->pos_old, 3*(size_t) pl->cap*sizeof (double));
    pl->id = realloc (pl->id, (size_t) pl->cap*sizeof (unsigned));
    if (pl->particulate) {
      pl->vel = realloc (pl->vel, 3*(size_t) pl->cap*sizeof (double));
      pl->force = realloc (pl->force, 3*(size_t) pl->cap*sizeof (double));
      pl->mass = realloc (pl->mass, (size_t) pl->cap*sizeof (double));
      pl->volume = realloc (pl->volume, (size_t) pl->cap*sizeof (double));
    }
  }
  for (int q = 0; q < n; q++) {
    const double * r = rec + (size_t) rs*q;
    int m = pl->n + q;
    memcpy (pl->pos + 3*m, r, 3*sizeof (double));
    memcpy (pl->pos_old + 3*m, r + 3, 3*sizeof (double));
    pl->id[m] = (unsigned) r[6];
    if (pl->particulate) {
      memcpy (pl->vel + 3*m, r + 7, 3*sizeof (double));
      pl->mass[m] = r[10];
      pl->volume[m] = r[11];
      memcpy (pl->force + 3*m, r + 12, 3*sizeof (double));
    }
  }
  pl->n += n;
}

int go_particles_count (const GoParticles * pl) { return pl->n; }
double * go_particles_pos (GoParticles * pl) { return pl->pos; }
unsigned * go_particles_id (GoParticles * pl) { return pl->id; }

static void move_particle (GoParticles * pl, int m, int q)
{
  memcpy (pl->pos + 3*m, pl->pos + 3*q, 3*sizeof (double));
  memcpy (pl->pos_old + 3*m, pl->pos_old + 3*q, 3*sizeof (double));
  pl->id[m] = pl->id[q];
  if (pl->particulate) {
    memcpy (pl->vel + 3*m, pl->vel + 3*q, 3*sizeof (double));
    memcpy (pl->force + 3*m, pl->force + 3*q, 3*sizeof (double));
    pl->mass[m] = pl->mass[q];
    pl->volume[m] = pl->volume[q];
  }
}

/* ---------------------------------------------------------------------------------------------
 * GfsParticulate with forces (modules/particulatecommon.c:91-842).  Uniform box, alpha = NULL
 * (fluid_rho = 1.), constant viscosity (GfsSourceDiffusion on U), no user coefficient functions
 * (cm = cl = 0.5, the default drag law).
 * ------------------------------------------------------------------------------------------- */

void go_particles_set_particulate (GoParticles * pl, const double * vel, const double * mass,
				   const double * volume)
{
  size_t np = (size_t) pl->n, cap = (size_t) (pl->cap > 0 ? pl->cap : 1);
  pl->particulate = 1;
  pl->vel = malloc (3*cap*sizeof (double));
  pl->force = calloc (3*cap, sizeof (double));
  pl->mass = malloc (cap*sizeof (double));
  pl->volume = malloc (cap*sizeof (double));
  memcpy (pl->vel, vel, 3*np*sizeof (double));
  memcpy (pl->mass, mass, np*sizeof (double));
  memcpy (pl->volume, volume, np*sizeof (double));
}

/* store_domain_previous_vel, particulatecommon.c:100-114: copy on the leaves, then the BC of the
 * new variable (a scalar with the default BC) */
static void store_previous_vel (GoSim * s, GoParticles * pl)
{
  GoDomain * dom = s->dom;
  int L = dom->depth;
  for (int c = 0; c < dom->dim; c++) {
    double * un = pl->uold[c]->lev[L];
    const double * u = s->u[c]->lev[L];
    const int * order = go_order (dom, L);
    size_t ncells = 1;
    for (int q = 0; q < dom->dim; q++) ncells *= (size_t) dom->n[L];
    for (size_t q = 0; q < ncells; q++)
      un[order[q]] = u[order[q]];
    go_bc (pl->uold[c], pl->uold[c], L);
  }
}

/* the list's forces, in the order they are applied (gfs_particle_list_read :1034-1070); every
 * GfsForceCoeff creates Un, Vn, Wn and stores the velocity when it is read (:181-187) */
void go_particles_set_forces (GoParticles * pl, GoSim * s, int n, const int * kinds)
{
  assert (n <= 8);
  pl->nforces = n;
  int coeff = 0;
  for (int q = 0; q < n; q++) {
    pl->forces[q] = kinds[q];
    if (kinds[q] != GO_FORCE_BUOY) coeff = 1;
  }
  if (coeff && !pl->uold[0]) {
    for (int c = 0; c < s->dom->dim; c++)
      pl->uold[c] = go_field_new (s->dom, -1);
    store_previous_vel (s, pl);
  }
}

void go_particles_set_coefficient (GoParticles * pl, int force,
				   double (* fn) (double, double, double, double, double))
{
  assert (force >= 0 && force < 8);
  pl->coefficient[force] = fn;
}

void go_particles_set_gravity (GoParticles * pl, const double g[3])
{
  for (int c = 0; c < 3; c++) pl->gravity[c] = g[c];
}

double * go_particles_vel (GoParticles * pl) { return pl->vel; }
double * go_particles_mass (GoParticles * pl) { return pl->mass; }
double * go_particles_force (GoParticles * pl) { return pl->force; }

/* gfs_center_gradient, fluid.c:434-475, both neighbours at the same level (x1 = x2 = 1.) */
static double center_gradient (const GoDomain * dom, const double * v, size_t cell, int c)
{
  int L = dom->depth;
  double v0 = v[cell];
  double x1 = 1., v1 = v[cell + dom->off[L][2*c + 1]];
  double x2 = 1., v2 = v[cell + dom->off[L][2*c]];
  return (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
}

/* compute_inertial_force, :285-336 */
static void inertial_force (GoSim * s, GoParticles * pl, const int cell[3], const double p[3],
			    double force[3])
{
  GoDomain * dom = s->dom;
  int dim = dom->dim, L = dom->depth;
  for (int c = 0; c < 3; c++) force[c] = 0;
  double size = 1./dom->n[L];
  double fluid_rho = 1.;
  double fluid_vel[3], fluid_veln[3];
  for (int c = 0; c < dim; c++)
    fluid_vel[c] = go_interpolate (dom, s->u[c]->lev[L], cell, p);
  for (int c = 0; c < dim; c++)
    fluid_veln[c] = go_interpolate (dom, pl->uold[c]->lev[L], cell, p);
  double dt = s->advection_params.dt;
  if (dt > 0.)
    for (int c = 0; c < dim; c++)
      force[c] = fluid_rho*(fluid_vel[c] - fluid_veln[c])/dt;
  else
    return;
  size_t idx = go_index (dom, L, cell[0], cell[1], dim == 3 ? cell[2] : 0);
  for (int c = 0; c < dim; c++)
    for (int c2 = 0; c2 < dim; c2++)
      force[c] += fluid_rho*center_gradient (dom, s->u[c]->lev[L], idx, c2)*
	s->u[c2]->lev[L][idx]/size;
}

/* vorticity_vector, :146-168 */
static void vorticity_vector (GoSim * s, size_t idx, double vort[3])
{
  GoDomain * dom = s->dom;
  int L = dom->depth;
  double size = 1./dom->n[L];
  const double * u = s->u[0]->lev[L], * v = s->u[1]->lev[L];
  if (dom->dim == 2) {
    vort[0] = 0.;
    vort[1] = 0.;
    vort[2] = (center_gradient (dom, v, idx, 0) - center_gradient (dom, u, idx, 1))/size;
  }
  else {
    const double * w = s->u[2]->lev[L];
    vort[0] = (center_gradient (dom, w, idx, 1) - center_gradient (dom, v, idx, 2))/size;
    vort[1] = (center_gradient (dom, u, idx, 2) - center_gradient (dom, w, idx, 0))/size;
    vort[2] = (center_gradient (dom, v, idx, 0) - center_gradient (dom, u, idx, 1))/size;
  }
}

/* the coefficient of a GfsForceCoeff with a GfsFunction: Rep, Urelp, Vrelp, Wrelp, Pdia set in the
 * cell, then gfs_function_value (compute_addedmass_force :354-384, compute_lift_force :462-485; the
 * relative velocity is an FttVector: all three components enter the norm) */
static double force_coefficient (GoSim * s, GoParticles * pl, int q, int f, const int cell[3],
				 const double p[3], double fluid_rho, double viscosity)
{
  GoDomain * dom = s->dom;
  int dim = dom->dim, L = dom->depth;
  const double * vel = pl->vel + 3*q;
  double relative_vel[3] = { 0. - vel[0], 0. - vel[1], 0. - vel[2] };
  for (int c = 0; c < dim; c++)
    relative_vel[c] = go_interpolate (dom, s->u[c]->lev[L], cell, p) - vel[c];
  double norm_relative_vel = sqrt (relative_vel[0]*relative_vel[0] +
				   relative_vel[1]*relative_vel[1] +
				   relative_vel[2]*relative_vel[2]);
  double dia = 2.*pow (3.0*pl->volume[q]/4.0/M_PI, 1./3.);
  if (viscosity == 0)
    viscosity = 0.001;
  double Re = norm_relative_vel*dia*fluid_rho/viscosity;
  return (* pl->coefficient[f]) (Re, relative_vel[0], relative_vel[1], relative_vel[2], dia);
}

/* gfs_particulate_event with forces, :768-842: forces from the state at the start of the step,
 * then pos += vel*dt/2, vel += force*dt/mass, pos += vel*dt/2 */
static void particulate_event (GoSim * s, GoParticles * pl, int q)
{
  GoDomain * dom = s->dom;
  int dim = dom->dim, L = dom->depth;
  double * p = pl->pos + 3*q, * vel = pl->vel + 3*q, * pf = pl->force + 3*q;
  double dt = s->advection_params.dt;
  double pos[3] = { p[0], p[1], p[2] };
  memcpy (pl->pos_old + 3*q, p, 3*sizeof (double));
  for (int c = 0; c < 3; c++) pf[c] = 0.;

  int cell[3];
  int inside = go_locate (dom, p, cell);
  size_t idx = inside ? go_index (dom, L, cell[0], cell[1], dim == 3 ? cell[2] : 0) : 0;
  double fluid_rho = 1.;
  double viscosity = s->visc[0];
  for (int f = 0; f < pl->nforces; f++) {
    double force[3] = { 0., 0., 0. };
    if (inside)
      switch (pl->forces[f]) {
      case GO_FORCE_INERTIAL:
	inertial_force (s, pl, cell, p, force);
	break;
      case GO_FORCE_ADDEDMASS: {   /* compute_addedmass_force, :363-427 */
	inertial_force (s, pl, cell, p, force);
	double cm = 0.5;
	if (pl->coefficient[f])
	  cm = force_coefficient (s, pl, q, f, cell, p, fluid_rho, viscosity);
	for (int c = 0; c < dim; c++)
	  force[c] *= cm;
	pl->mass[q] += fluid_rho*pl->volume[q]*cm;
	break;
      }
      case GO_FORCE_LIFT: {        /* compute_lift_force, :455-524 */
	double relative_vel[3] = { 0., 0., 0. }, vorticity[3];
	for (int c = 0; c < dim; c++)
	  relative_vel[c] = go_interpolate (dom, s->u[c]->lev[L], cell, p) - vel[c];
	if (dim == 2) relative_vel[2] = 0. - vel[2];
	vorticity_vector (s, idx, vorticity);
	double cl = 0.5;
	if (pl->coefficient[f])
	  cl = force_coefficient (s, pl, q, f, cell, p, fluid_rho, viscosity);
	if (dim == 2) {
	  force[0] = fluid_rho*cl*relative_vel[1]*vorticity[2];
	  force[1] = -fluid_rho*cl*relative_vel[0]*vorticity[2];
	}
	else {
	  force[0] = fluid_rho*cl*(relative_vel[1]*vorticity[2] - relative_vel[2]*vorticity[1]);
	  force[1] = fluid_rho*cl*(relative_vel[2]*vorticity[0] - relative_vel[0]*vorticity[2]);
	  force[2] = fluid_rho*cl*(relative_vel[0]*vorticity[1] - relative_vel[1]*vorticity[0]);
	}
	break;
      }
      case GO_FORCE_DRAG: {        /* compute_drag_force, :552-588 */
	double relative_vel[3] = { 0., 0., 0. };
	for (int c = 0; c < dim; c++)
	  relative_vel[c] = go_interpolate (dom, s->u[c]->lev[L], cell, p) - vel[c];
	double dia = 2.*pow (3.0*pl->volume[q]/4.0/M_PI, 1./3.);
	double norm_relative_vel = dim == 3 ?
	  sqrt (relative_vel[0]*relative_vel[0] + relative_vel[1]*relative_vel[1] +
		relative_vel[2]*relative_vel[2]) :
	  sqrt (relative_vel[0]*relative_vel[0] + relative_vel[1]*relative_vel[1]);
	double cd = 0., Re;
	if (viscosity == 0)
	  break;
	Re = norm_relative_vel*dia*fluid_rho/viscosity;
	if (pl->coefficient[f])      /* :565-574 */
	  cd = (* pl->coefficient[f]) (Re, relative_vel[0], relative_vel[1], relative_vel[2], dia);
	else if (Re < 1e-8)
	  break;
	else if (Re < 50.0)
	  cd = 16.*(1. + 0.15*pow (Re, 0.5))/Re;
	else
	  cd = 48.*(1. - 2.21/pow (Re, 0.5))/Re;
	for (int c = 0; c < dim; c++)
	  force[c] += 3./(4.*dia)*cd*norm_relative_vel*relative_vel[c]*fluid_rho;
	break;
      }
      case GO_FORCE_BUOY:          /* compute_buoyancy_force, :619-653 */
	for (int c = 0; c < dim; c++)
	  force[c] += (pl->mass[q]/pl->volume[q] - fluid_rho)*pl->gravity[c];
	break;
      }
    /* compute_forces, :738-752 */
    double total[3];
    for (int c = 0; c < dim; c++)
      total[c] = force[c]*pl->volume[q] + pf[c];
    if (dim == 2) total[2] = 0.;
    for (int c = 0; c < 3; c++) pf[c] = total[c];
  }
  for (int c = 0; c < dim; c++) {
    pos[c] += vel[c]*dt/2.;
    vel[c] += pf[c]*dt/pl->mass[q];
    pos[c] += vel[c]*dt/2.;
  }
  p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
}

/* gfs_particle_list_event, particulatecommon.c:980-1015 (no forces):
 *   remove_particles_not_in_domain (:955-969)
 *   every particle: gfs_particle_event (src/particle.c:31-44)
 *   gfs_particle_bc (:3375-3395): particles that left the box are wrapped through the periodic
 *   side they crossed (periodic_bc_particle :3189-3214); on non-periodic sides they stay where
 *   they are and are removed at the next event. */
void go_particle_list_event (GoSim * s, GoParticles * pl)
{
  GoDomain * dom = s->dom;
  double dt = s->advection_params.dt;
  int cell[3];
  /* remove particles not in domain (order of the survivors kept) */
  int m = 0;
  for (int q = 0; q < pl->n; q++)
    if (go_locate (dom, pl->pos + 3*q, cell)) {
      if (m != q)
	move_particle (pl, m, q);
      m++;
    }
  pl->n = m;
  for (int q = 0; q < pl->n; q++) {
    if (pl->particulate && pl->nforces > 0) {
      particulate_event (s, pl, q);     /* gfs_particulate_event, :768-842 */
      continue;
    }
    double * p = pl->pos + 3*q, * po = pl->pos_old + 3*q;
    double pos[3] = { p[0], p[1], p[2] };
    po[0] = p[0]; po[1] = p[1]; po[2] = p[2];
    go_advect_point (s, pos, dt);
    p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
  }
  /* boundary conditions: a particle that left the box is taken off the list
     (list_boundary_particles :3326-3359) and only a periodic side puts it back */
  char * drop = calloc (pl->n + 1, 1);
  for (int q = 0; q < pl->n; q++) {
    double * p = pl->pos + 3*q, * po = pl->pos_old + 3*q;
    if (go_locate (dom, p, cell))
      continue;
    int d = boundary_direction (dom, po, p);
    if (dom->side[d] == GO_SIDE_EXTERNAL) {
      /* GfsBoundaryMpi: the particle object is sent to the process across the side
	 (send_particles :3247-3268); its position is the same point of space, i.e. one box size
	 less along the normal in the coordinates of the receiving box */
      double normal = (double) (d ^ 1) - (double) d;
      int rs = go_particles_record_size (pl);
      pl->out[d] = realloc (pl->out[d], (size_t) rs*(pl->nout[d] + 1)*sizeof (double));
      double * r = pl->out[d] + (size_t) rs*pl->nout[d]++;
      for (int c = 0; c < 3; c++) { r[c] = p[c]; r[3 + c] = po[c]; }
      r[d/2] -= normal*1.;
      r[3 + d/2] -= normal*1.;
      r[6] = (double) pl->id[q];
      if (pl->particulate) {
	for (int c = 0; c < 3; c++) { r[7 + c] = pl->vel[3*q + c]; r[12 + c] = pl->force[3*q + c]; }
	r[10] = pl->mass[q];
	r[11] = pl->volume[q];
      }
      drop[q] = 1;
    }
    else if (dom->side[d] != GO_SIDE_PERIODIC)
      drop[q] = 1;
    else {
      /* periodic_bc_particle: box of size 1 centred on the origin, matching box = itself */
      double size = 1.;
      double normal = (double) (d ^ 1) - (double) d;
      double box_face = 0., box_face_nbr = 0.;
      box_face += normal*size/2.;
      box_face_nbr -= normal*size/2.;
      double tolerance = size/1.e8;
      double distance = (p[d/2] - box_face)*normal;
      p[d/2] = box_face_nbr + distance + normal*tolerance;
      po[d/2] = p[d/2];
    }
  }
  m = 0;
  for (int q = 0; q < pl->n; q++)
    if (!drop[q]) {
      if (m != q)
	move_particle (pl, m, q);
      m++;
    }
  pl->n = m;
  free (drop);
  /* the velocity of this step for the inertial force of the next one (:1003-1012: only for
     GfsForceInertial objects) */
  for (int f = 0; f < pl->nforces; f++)
    if (pl->forces[f] == GO_FORCE_INERTIAL) {
      store_previous_vel (s, pl);
      break;
    }
}
