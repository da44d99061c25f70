// simulation.hip -- host control of the projection + advection time step: the body of
// simulation_run (src/simulation.c:432-557) and the drivers of src/timestep.c, statement by
// statement; every traversal is a kernel launch on the domain's stream.
#include "gfship_internal.hpp"
#include <cmath>
#include <cfloat>
#include <cstdlib>

using namespace gfship;

#define G_MAXINT 2147483647

struct gfship_sim {
  gfship_domain * dom = nullptr;
  gfship_field p = -1, pmac = -1, u[3] = {-1, -1, -1}, g[3] = {-1, -1, -1}, gmac[3] = {-1, -1, -1};
  std::vector<gfship_field> tracers;
  std::vector<int> tracer_gradient;    // GfsVariableTracer { gradient = }: 0 centred, 1 van Leer
  gfship_field un[3] = {-1, -1, -1};   // face normal velocities (see timestep_kernels.hip)
  gfship_field fv[6] = {-1, -1, -1, -1, -1, -1};
  gfship_field dia = -1, div = -1, res = -1, tmp = -1; // temporaries of mac_projection
  gfship_multilevel_params projection_params, approx_projection_params;
  gfship_advection_params advection_params;
  double t = 0., end = DBL_MAX, dtmax = DBL_MAX, tnext = 0.;
  unsigned i = 0, iend = G_MAXINT;
  double visc[3] = {0., 0., 0.};   // GfsSourceDiffusion on U, V, W (constant coefficient)
  // GfsPhysicalParams { alpha = ... }: gfs_function_face_value (alpha) on the faces normal to c, in the
  // layout gfship_poisson_coefficients_alpha takes (gfship_sim_set_alpha); has_alpha = false: alpha = NULL
  gfship_field alpha[3] = {-1, -1, -1};
  bool has_alpha = false;
  gfship_multilevel_params diffusion_params[3];
  gfship_field drhs = -1, rhoc = -1;   // temporaries of variable_diffusion
  bool cfl_ready = false;              // maxima for the CFL condition left by the last projection
  bool div_ready = false;              // div holds the scaled divergence of un for the MAC projection with div_dt
  double div_dt = 0.;
  // the corrected MAC velocities of the last approximate projection have not been stored (nobody reads
  // them before the next predictor overwrites them: project_correct_lazy_kernel); materialize_un
  // rebuilds them from the uncorrected centred velocities (the leaf storage swapped out into adv_tmp3)
  // and p when somebody asks
  bool un_lazy = false;
  double un_lazy_dt = 0.;
  bool un_handle_given = false;        // the caller holds the field handles of un: no lazy path any more
  gfship_field adv_tmp = -1;           // output of the fused advection kernel (swapped with v)
  gfship_field adv_tmp3[3] = {-1, -1, -1};   // the same for the three-component kernel
  gfship_next_event_fn next_event = nullptr; void * next_event_ctx = nullptr;
};

namespace {

inline double * leaf (gfship_sim * s, gfship_field f)
{
  return s->dom->fields[f].lev[s->dom->depth];
}

void ptrs3 (gfship_sim * s, const gfship_field f[3], double * out[3])
{
  for (int c = 0; c < 3; c++)
    out[c] = (c < s->dom->dim && f[c] >= 0) ? leaf (s, f[c]) : nullptr;
}

void ptrs6 (gfship_sim * s, double * out[6])
{
  for (int d = 0; d < 6; d++)
    out[d] = d < 2*s->dom->dim ? leaf (s, s->fv[d]) : nullptr;
}

#define TRY(x) do { int r_ = (x); if (r_ != GFSHIP_OK) return r_; } while (0)

int bc_leaf (gfship_sim * s, gfship_field v)
{
  Field * V = get_field (s->dom, v);
  return launch_bc (s->dom, V, V, s->dom->depth, 0);
}

// gfs_domain_bc of the components of a vector, one after the other in the reference: one launch,
// one message per MPI side
int bc_leaf_vector (gfship_sim * s, const gfship_field v[3])
{
  Field * V[3];
  for (int c = 0; c < s->dom->dim; c++)
    if (!(V[c] = get_field (s->dom, v[c]))) return GFSHIP_EINVAL;
  return launch_bc_multi (s->dom, V, s->dom->dim, s->dom->depth, 0);
}

// the MAC velocities the last approximate projection left unstored (gfship_sim.un_lazy):
// gfs_face_interpolated_normal_velocity of the uncorrected velocities + gfs_correct_normal_velocities,
// the kernels the unfused path runs (same expressions as the fused ones: same bits)
int materialize_un (gfship_sim * s)
{
  if (!s->un_lazy) return GFSHIP_OK;
  s->un_lazy = false;
  double * uold[3], * un[3];
  ptrs3 (s, s->adv_tmp3, uold);
  ptrs3 (s, s->un, un);
  TRY (launch_face_interp_un (s->dom, uold, un));
  TRY (launch_correct_un (s->dom, leaf (s, s->p), un, s->un_lazy_dt));
  return GFSHIP_OK;
}

// mac_projection, src/timestep.c:356-444.  `pdata` supplies the storage of the pressure and
// `pbc` the boundary conditions (gfs_variables_swap swaps storage only, src/variable.c:234-243).
// lazy: the approximate projection of the loop body of simulation_run, whose corrected MAC velocities
// nobody reads (no tracers, no handle given out): they are not stored (see gfship_sim.un_lazy)
int mac_projection (gfship_sim * s, gfship_multilevel_params * par, double dt, gfship_field p,
		    const gfship_field g[3], bool approximate = false, bool lazy = false)
{
  gfship_domain * dom = s->dom;
  double * un[3], * gp[3], * u[3];
  ptrs3 (s, s->un, un);
  ptrs3 (s, g, gp);
  ptrs3 (s, s->u, u);
  /* gfs_reset_gradients + no face sources: g is overwritten by centered_gradient below */
  /* gfs_poisson_coefficients: unit weights for alpha = NULL, else the face weights of every level */
  if (s->has_alpha) {
    TRY (gfship_poisson_coefficients_alpha (dom, s->alpha));
    lazy = false;
  }
  else
    TRY (gfship_poisson_coefficients (dom));
  /* dia = 0 on all levels (gfs_cell_reset on FTT_TRAVERSE_ALL) */
  for (int l = 0; l <= dom->depth; l++)
    if (!dom->fields[s->dia].zero[l])
      TRY (gfship_field_fill (dom, s->dia, l, 0.));
  /* MAC divergence, scaled by 1/dt (approximate projection: together with the MAC velocities
     interpolated from the centred ones, gfs_approximate_projection src/timestep.c:572-580) */
  for (int c = 0; c < dom->dim; c++)
    if (s->visc[c] != 0. || dom->src[c] != 0.) lazy = false;   /* the CFL needs the full kernel then */
  if (!approximate || !s->tracers.empty () || s->un_handle_given || p != s->p || getenv ("GFSHIP_NO_LAZY_UN"))
    lazy = false;
  if (!approximate)
    TRY (materialize_un (s));          /* un is an input of the MAC projection */
  else
    s->un_lazy = false;                /* un is rewritten (or declared unstored) below */
  if (approximate)
    TRY (launch_face_interp_div (dom, u, lazy ? nullptr : un, leaf (s, s->div), dt));
  else if (!(s->div_ready && s->div_dt == dt))       /* left there by the predictor (sweep along z) */
    TRY (launch_divergence (dom, un, leaf (s, s->div), dt));
  s->div_ready = false;
  TRY (gfship_poisson_solve (dom, par, p, s->div, s->res, s->dia, dt));
  /* gfs_correct_normal_velocities + gfs_scale_gradients, and for the approximate projection
     gfs_correct_centered_velocities (src/timestep.c:486-530), in one pass over p.  The pass
     also leaves the largest |un|, |u| behind for the CFL condition of the next time step. */
  s->cfl_ready = false;
  bool want_max = approximate;
  for (int c = 0; c < dom->dim; c++)
    if (s->visc[c] != 0. || dom->src[c] != 0.) want_max = false;   /* the acceleration term needs the full kernel */
  if (s->has_alpha) {
    /* the same pass with gfs_face_weighted_gradient's weights (the CFL maxima come from their own pass) */
    want_max = false;
    TRY (launch_project_correct_weighted (dom, leaf (s, p), un, gp, approximate ? u : nullptr, dt));
  }
  else if (lazy) {
    /* g and the corrected centred velocities (out of place: the faces read uncorrected neighbours), the
       maxima of |un|, |u| for the CFL condition; the storage of U, V, W is swapped with the scratch */
    const int L = dom->depth;
    double * uo[3];
    for (int c = 0; c < dom->dim; c++) {
      if (s->adv_tmp3[c] < 0)
	s->adv_tmp3[c] = gfship_field_alloc (dom, -1);
      if (s->adv_tmp3[c] < 0) return s->adv_tmp3[c];
    }
    ptrs3 (s, s->adv_tmp3, uo);
    TRY (launch_project_correct_lazy (dom, leaf (s, p), u, gp, uo, dt));
    for (int c = 0; c < dom->dim; c++) {
      std::swap (dom->fields[s->u[c]].lev[L], dom->fields[s->adv_tmp3[c]].lev[L]);
      dom->fields[s->u[c]].zero[L] = false;
    }
    s->un_lazy = true;
    s->un_lazy_dt = dt;
  }
  else
    TRY (launch_project_correct (dom, leaf (s, p), un, gp, approximate ? u : nullptr, dt, want_max));
  s->cfl_ready = want_max;
  TRY (bc_leaf_vector (s, g));
  if (approximate)
    TRY (bc_leaf_vector (s, s->u));
  return GFSHIP_OK;
}

// gfs_correct_centered_velocities, src/timestep.c:498-530.  with_coarse: the caller's next operation is
// gfs_cell_coarse_init: the level below the leaves of U, V, W is then filled by the same pass (3-D)
int correct_centered_velocities (gfship_sim * s, const gfship_field g[3], double dt,
				 bool * with_coarse = nullptr)
{
  gfship_domain * dom = s->dom;
  double * u[3], * gp[3];
  ptrs3 (s, s->u, u);
  ptrs3 (s, g, gp);
  if (with_coarse && dom->dim == 3 && dom->depth >= 1) {
    double * uc[3];
    for (int c = 0; c < 3; c++) {
      Field * F = get_field (dom, s->u[c]);
      uc[c] = F->lev[dom->depth - 1];
      F->zero[dom->depth - 1] = false;
    }
    TRY (launch_correct_centered_coarse (dom, u, gp, dt, uc));
    *with_coarse = true;
  }
  else {
    if (with_coarse) *with_coarse = false;
    TRY (launch_correct_centered (dom, u, gp, dt));
  }
  TRY (bc_leaf_vector (s, s->u));
  return GFSHIP_OK;
}

// face_values_set, src/timestep.c:644-654
int face_values_set (gfship_sim * s, gfship_field v, double dt, int use_centered, int gradient,
		     int cmask = 7)
{
  double * u[3], * un[3], * fv[6];
  ptrs3 (s, s->u, u);
  ptrs3 (s, s->un, un);
  ptrs6 (s, fv);
  /* v->sources: the implicit diffusion of a velocity component acts as MAC source */
  double visc = 0., gsrc = 0.;
  for (int c = 0; c < s->dom->dim; c++)
    if (v == s->u[c]) {
      visc = s->visc[c];
      gsrc = s->dom->src[c];     /* ... and the intensity of a GfsSource on it */
    }
  TRY (launch_advected_face_values (s->dom, leaf (s, v), u, un, dt, use_centered, gradient, fv, cmask,
				    visc, gsrc));
  TRY (launch_face_bc (s->dom, get_field (s->dom, v), fv, cmask));
  return GFSHIP_OK;
}

// variable_sources, src/timestep.c:872-921 (Godunov; the update goes into sv)
int variable_sources (gfship_sim * s, gfship_field v, gfship_field sv, int gradient, bool velocity,
		      double dt, const gfship_field gmac[3], const gfship_field g[3])
{
  double * un[3], * fv[6];
  ptrs3 (s, s->un, un);
  ptrs6 (s, fv);
  int c = s->dom->fields[v].component;
  if (godunov_fused_supported (s->dom) && gradient <= 1) {      /* limiters 2 .. 4: the general path */
    /* periodic box: face values recomputed inside the flux kernel, no face-value arrays */
    gfship_domain * dom = s->dom;
    const int L = dom->depth;
    const double * gm = velocity ? leaf (s, gmac[c]) : nullptr;
    const double * gc = (velocity && g) ? leaf (s, g[c]) : nullptr;
    double visc = 0., gsrc = 0.;
    for (int q = 0; q < dom->dim; q++)
      if (v == s->u[q]) {
	visc = s->visc[q];
	gsrc = dom->src[q];
      }
    dom->fields[sv].zero[L] = false;
    if (sv != v)      /* sv holds a copy of v (source_diffusion): out = sv */
      return launch_advect_fused (dom, velocity, leaf (s, v), leaf (s, sv), un, gm, gc, dt,
				  gradient, visc, gsrc);
    /* in place in the reference: here into a scratch leaf level, then the storage is swapped
       (every cell reads the old values of its neighbours) */
    if (s->adv_tmp < 0)
      s->adv_tmp = gfship_field_alloc (dom, -1);
    if (s->adv_tmp < 0) return s->adv_tmp;
    TRY (launch_advect_fused (dom, velocity, leaf (s, v), leaf (s, s->adv_tmp), un, gm, gc, dt,
			      gradient, visc, gsrc));
    std::swap (dom->fields[v].lev[L], dom->fields[s->adv_tmp].lev[L]);
    return GFSHIP_OK;
  }
  TRY (face_values_set (s, v, dt, 0, gradient));
  const double * gm = velocity ? leaf (s, gmac[c]) : nullptr;
  const double * gc = (velocity && g) ? leaf (s, g[c]) : nullptr;
  s->dom->fields[sv].zero[s->dom->depth] = false;
  double gsrc = 0.;
  for (int q = 0; q < s->dom->dim; q++)
    if (v == s->u[q])
      gsrc = s->dom->src[q];
  TRY (launch_flux_update (s->dom, velocity, leaf (s, sv), un, fv, gm, gc, dt, gsrc));
  return GFSHIP_OK;
}

// variable_diffusion, src/timestep.c:923-949
int variable_diffusion (gfship_sim * s, int c, gfship_field rhs)
{
  gfship_domain * dom = s->dom;
  gfship_multilevel_params * par = &s->diffusion_params[c];
  double dt = s->advection_params.dt;
  if (s->rhoc < 0)
    s->rhoc = gfship_field_alloc (dom, -1);
  if (s->rhoc < 0) return s->rhoc;
  TRY (gfship_diffusion_coefficients (dom, s->visc[c], dt, s->rhoc, par->beta));
  TRY (gfship_diffusion_rhs (dom, s->u[c], rhs, s->rhoc, par->beta));
  TRY (gfship_diffusion (dom, par, s->u[c], rhs, s->rhoc));
  return GFSHIP_OK;
}

int advance_tracers (gfship_sim * s, double dt)
{
  if (!s->tracers.empty ())
    TRY (materialize_un (s));
  for (gfship_field t : s->tracers)
    TRY (gfship_tracer_advection (s, t, dt));
  return GFSHIP_OK;
}

} // namespace

extern "C" {

/* internal view of a simulation for particles.hip */
struct gfship_sim_view { gfship_domain * dom; const gfship_field * u; double dt; double visc; };
gfship_sim_view gfship_sim_view_get (gfship_sim * s)
{
  gfship_sim_view v = { s->dom, s->u, s->advection_params.dt, s->visc[0] };
  return v;
}

int gfship_sim_create (gfship_sim ** out, gfship_domain * dom)
{
  GFSHIP_CHECK (out && dom, GFSHIP_EINVAL, "null argument");
  *out = nullptr;
  gfship_sim * s = new gfship_sim;
  s->dom = dom;
  auto alloc = [&] (int comp) { return gfship_field_alloc (dom, comp); };
  s->p = alloc (-1);
  s->pmac = alloc (-1);
  for (int c = 0; c < dom->dim; c++) {
    s->u[c] = alloc (c);
    s->g[c] = alloc (c);      /* gfs_variable_set_vector (g / gmac), src/simulation.c:455-456 */
    s->gmac[c] = alloc (c);
    s->un[c] = alloc (-1);
  }
  for (int d = 0; d < 2*dom->dim; d++)
    s->fv[d] = alloc (-1);
  s->dia = alloc (-1);
  s->div = alloc (-1);
  s->res = alloc (-1);
  gfship_multilevel_params_init (&s->projection_params, dom->dim);
  gfship_multilevel_params_init (&s->approx_projection_params, dom->dim);
  for (int c = 0; c < 3; c++) { /* diffusion_init, src/source.c:966-974 */
    gfship_multilevel_params_init (&s->diffusion_params[c], dom->dim);
    s->diffusion_params[c].tolerance = 1e-6;
  }
  /* gfs_advection_params_init, src/advection.c:922-942 */
  s->advection_params.cfl = 0.8;
  s->advection_params.dt = 0.;
  s->advection_params.gradient = 0;
  s->advection_params.gc = 1;
  if (s->res < 0) {
    gfship_sim_destroy (s);
    return GFSHIP_ENOMEM;
  }
  /* the MAC velocities a step leaves unstored are stored before anybody overwrites a field */
  dom->before_write = [] (void * ctx) -> int { return materialize_un ((gfship_sim *) ctx); };
  dom->before_write_ctx = s;
  *out = s;
  return GFSHIP_OK;
}

void gfship_sim_destroy (gfship_sim * s)
{
  if (!s) return;
  gfship_domain * dom = s->dom;
  if (dom->before_write_ctx == s) { dom->before_write = nullptr; dom->before_write_ctx = nullptr; }
  auto fr = [&] (gfship_field f) { if (f >= 0) gfship_field_free (dom, f); };
  fr (s->p); fr (s->pmac);
  for (int c = 0; c < 3; c++) { fr (s->u[c]); fr (s->g[c]); fr (s->gmac[c]); fr (s->un[c]); }
  for (int d = 0; d < 6; d++) fr (s->fv[d]);
  fr (s->dia); fr (s->div); fr (s->res); fr (s->drhs); fr (s->rhoc); fr (s->adv_tmp);
  for (int c = 0; c < 3; c++) fr (s->adv_tmp3[c]);
  for (gfship_field t : s->tracers) fr (t);
  delete s;
}

gfship_field gfship_sim_variable (gfship_sim * s, int which, int c)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  GFSHIP_CHECK (c >= 0 && (which == GFSHIP_VAR_TRACER || c < s->dom->dim), GFSHIP_EINVAL,
		"component %d out of range", c);
  switch (which) {
  case GFSHIP_VAR_P: return s->p;
  case GFSHIP_VAR_PMAC: return s->pmac;
  case GFSHIP_VAR_U: return s->u[c];
  case GFSHIP_VAR_G: return s->g[c];
  case GFSHIP_VAR_GMAC: return s->gmac[c];
  case GFSHIP_VAR_TRACER:
    GFSHIP_CHECK ((size_t) c < s->tracers.size (), GFSHIP_EINVAL, "no tracer %d", c);
    return s->tracers[c];
  case GFSHIP_VAR_UN:
    if (materialize_un (s) != GFSHIP_OK) return GFSHIP_EHIP;
    s->un_handle_given = true;
    return s->un[c];
  }
  set_error ("unknown variable kind %d", which);
  return GFSHIP_EINVAL;
}

gfship_multilevel_params * gfship_sim_projection_params (gfship_sim * s)
{ return s ? &s->projection_params : nullptr; }
gfship_multilevel_params * gfship_sim_approx_projection_params (gfship_sim * s)
{ return s ? &s->approx_projection_params : nullptr; }
gfship_advection_params * gfship_sim_advection_params (gfship_sim * s)
{ return s ? &s->advection_params : nullptr; }

int gfship_sim_set_time (gfship_sim * s, double end, double dtmax)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  s->end = end;
  s->dtmax = dtmax;
  return GFSHIP_OK;
}

double gfship_sim_time (gfship_sim * s) { return s ? s->t : 0.; }
unsigned gfship_sim_iter (gfship_sim * s) { return s ? s->i : 0; }

int gfship_sim_set_source (gfship_sim * s, int c, double intensity)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  GFSHIP_CHECK (c >= 0 && c < s->dom->dim, GFSHIP_EINVAL, "component %d out of range", c);
  s->dom->src[c] = intensity;
  s->cfl_ready = false;
  return GFSHIP_OK;
}

int gfship_sim_set_alpha (gfship_sim * s, const gfship_field alpha[3])
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  if (!alpha) {
    s->has_alpha = false;
    return GFSHIP_OK;
  }
  for (int c = 0; c < s->dom->dim; c++) {
    GFSHIP_CHECK (get_field (s->dom, alpha[c]) != nullptr, GFSHIP_EINVAL, "alpha[%d] is not a field of the domain", c);
    s->alpha[c] = alpha[c];
  }
  for (int c = 0; c < s->dom->dim; c++)
    GFSHIP_CHECK (s->visc[c] == 0., GFSHIP_EUNSUPPORTED,
		  "GfsSourceDiffusion together with GfsPhysicalParams { alpha } (gfs_diffusion_coefficients "
		  "with a variable density, src/poisson.c:1280-1348) is not supported");
  s->has_alpha = true;
  s->cfl_ready = false;
  return GFSHIP_OK;
}

int gfship_sim_set_viscosity (gfship_sim * s, int c, double nu)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  GFSHIP_CHECK (c >= 0 && c < s->dom->dim, GFSHIP_EINVAL, "component %d out of range", c);
  GFSHIP_CHECK (nu >= 0., GFSHIP_EINVAL, "the diffusion coefficient must be positive");
  GFSHIP_CHECK (nu == 0. || !s->has_alpha, GFSHIP_EUNSUPPORTED,
		"GfsSourceDiffusion together with GfsPhysicalParams { alpha } is not supported");
  s->visc[c] = nu;
  return GFSHIP_OK;
}

gfship_multilevel_params * gfship_sim_diffusion_params (gfship_sim * s, int c)
{ return (s && c >= 0 && c < 3) ? &s->diffusion_params[c] : nullptr; }

int gfship_sim_set_next_event (gfship_sim * s, gfship_next_event_fn fn, void * ctx)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  s->next_event = fn;
  s->next_event_ctx = ctx;
  return GFSHIP_OK;
}

int gfship_sim_add_tracer (gfship_sim * s)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  gfship_field t = gfship_field_alloc (s->dom, -1);
  if (t < 0) return t;
  s->tracers.push_back (t);
  s->tracer_gradient.push_back (1);      /* gfs_center_van_leer_gradient, src/variable.c:429 */
  return (int) s->tracers.size () - 1;
}

// mac_dt != 0: the loop body of simulation_run, where the MAC projection with time step mac_dt follows
static int predicted_face_velocities (gfship_sim * s, double mac_dt);

int gfship_predicted_face_velocities (gfship_sim * s)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  return predicted_face_velocities (s, 0.);
}

static int predicted_face_velocities (gfship_sim * s, double mac_dt)
{
  double * fv[6];
  ptrs6 (s, fv);
  s->div_ready = false;
  s->un_lazy = false;                  /* every face is rewritten below */
  /* the reset of every face (gfs_face_reset_normal_velocity) is implied: each component's
     faces are all overwritten below */
  if ((godunov_fused_supported (s->dom) || godunov_fused_mpi_supported (s->dom)) &&
      s->advection_params.gradient <= 1) {
    double * u[3], * un[3];
    ptrs3 (s, s->u, u);
    ptrs3 (s, s->un, un);
    const bool want_div = mac_dt != 0. && s->dom->dim == 3 && !getenv ("GFSHIP_NO_FUSED_DIVERGENCE");
    bool done = false;
    TRY (launch_predict_un_fused (s->dom, u, s->advection_params.dt,
				  s->advection_params.gradient, s->visc, un,
				  want_div ? leaf (s, s->div) : nullptr, mac_dt, &done));
    if (done) {
      s->div_ready = true;
      s->div_dt = mac_dt;
      s->dom->fields[s->div].zero[s->dom->depth] = false;
    }
    return GFSHIP_OK;
  }
  for (int c = 0; c < s->dom->dim; c++) {
    /* only the faces normal to component c are read by gfs_face_advected_normal_velocity */
    TRY (face_values_set (s, s->u[c], s->advection_params.dt, 1, s->advection_params.gradient, 1 << c));
    TRY (launch_predict_un (s->dom, c, leaf (s, s->u[c]), fv, leaf (s, s->un[c])));
  }
  return GFSHIP_OK;
}

int gfship_mac_projection (gfship_sim * s, gfship_multilevel_params * par, double dt,
			   gfship_field p, const gfship_field g[3])
{
  GFSHIP_CHECK (s && par && g, GFSHIP_EINVAL, "null argument");
  return mac_projection (s, par, dt, p, g);
}

int gfship_approximate_projection (gfship_sim * s, gfship_multilevel_params * par, double dt,
				   gfship_field p, const gfship_field g[3])
{
  GFSHIP_CHECK (s && par && g, GFSHIP_EINVAL, "null argument");
  /* MAC velocities from the centred ones, projection, correction of the centred velocities */
  TRY (mac_projection (s, par, dt, p, g, true));
  return GFSHIP_OK;
}

// corr_dt != 0 (the loop body of simulation_run): the caller's next operation is
// gfs_correct_centered_velocities (g, corr_dt) followed by gfs_cell_coarse_init -- where the three
// components are advected in one pass that pass also applies the correction and fills the level below
// the leaves of U, V, W (*corrected, *u_coarse set)
static int centered_velocity_advection (gfship_sim * s, const gfship_field gmac[3], const gfship_field g[3],
					 double corr_dt, bool * corrected, bool * u_coarse);

int gfship_centered_velocity_advection (gfship_sim * s, const gfship_field gmac[3],
					const gfship_field g[3])
{
  GFSHIP_CHECK (s && gmac, GFSHIP_EINVAL, "null argument");
  return centered_velocity_advection (s, gmac, g, 0., nullptr, nullptr);
}

static int centered_velocity_advection (gfship_sim * s, const gfship_field gmac[3], const gfship_field g[3],
					 double corr_dt, bool * corrected, bool * u_coarse)
{
  if (corrected) *corrected = false;
  if (u_coarse) *u_coarse = false;
  if (s->dom->dim == 3 && (godunov_fused_supported (s->dom) || godunov_fused_mpi_supported (s->dom)) &&
      !s->dom->no_fused_godunov3 && s->advection_params.gradient <= 1 &&
      s->visc[0] == 0. && s->visc[1] == 0. && s->visc[2] == 0.) {
    /* the three components in one pass over the box (same MAC velocities, nothing of one component
       feeds another): into scratch leaf levels, then the storage is swapped */
    gfship_domain * dom = s->dom;
    const int L = dom->depth;
    double * v[3], * out[3], * un[3], * gm[3], * gc[3];
    for (int c = 0; c < 3; c++) {
      if (s->adv_tmp3[c] < 0)
	s->adv_tmp3[c] = gfship_field_alloc (dom, -1);
      if (s->adv_tmp3[c] < 0) return s->adv_tmp3[c];
    }
    ptrs3 (s, s->u, v);
    ptrs3 (s, s->adv_tmp3, out);
    ptrs3 (s, s->un, un);
    ptrs3 (s, gmac, gm);
    if (g) ptrs3 (s, g, gc);
    const bool fuse = corr_dt != 0. && g && corrected && !getenv ("GFSHIP_NO_FUSED_CORRECTION");
    double * uc[3] = { nullptr, nullptr, nullptr };
    if (fuse && u_coarse && L >= 1)
      for (int c = 0; c < 3; c++) {
	Field * F = get_field (dom, s->u[c]);
	uc[c] = F->lev[L - 1];
	F->zero[L - 1] = false;
      }
    TRY (launch_advect3_fused (dom, v, out, un, gm, g ? gc : nullptr, s->advection_params.dt,
			       s->advection_params.gradient, fuse ? corr_dt : 0., uc[0] ? uc : nullptr));
    for (int c = 0; c < 3; c++) {
      std::swap (dom->fields[s->u[c]].lev[L], dom->fields[s->adv_tmp3[c]].lev[L]);
      dom->fields[s->u[c]].zero[L] = false;
    }
    if (fuse) {
      *corrected = true;
      if (u_coarse) *u_coarse = uc[0] != nullptr;
    }
    TRY (bc_leaf_vector (s, s->u));
    return GFSHIP_OK;
  }
  for (int c = 0; c < s->dom->dim; c++) {
    if (s->visc[c] != 0.) {
      /* source_diffusion (v[c]): rhs = copy of v on the leaves, sources into rhs, implicit
	 solve (src/timestep.c:996-1007) */
      gfship_domain * dom = s->dom;
      if (s->drhs < 0)
	s->drhs = gfship_field_alloc (dom, -1);
      if (s->drhs < 0) return s->drhs;
      GFSHIP_HIP (hipMemcpyAsync (leaf (s, s->drhs), leaf (s, s->u[c]),
				  dom->lay[dom->depth].total*sizeof (double),
				  hipMemcpyDeviceToDevice, dom->stream));
      TRY (variable_sources (s, s->u[c], s->drhs, s->advection_params.gradient, true,
			     s->advection_params.dt, gmac, g));
      TRY (variable_diffusion (s, c, s->drhs));
    }
    else
      TRY (variable_sources (s, s->u[c], s->u[c], s->advection_params.gradient, true,
			     s->advection_params.dt, gmac, g));
  }
  TRY (bc_leaf_vector (s, s->u));
  return GFSHIP_OK;
}

int gfship_tracer_advection (gfship_sim * s, gfship_field t, double dt)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  TRY (materialize_un (s));
  if (!get_field (s->dom, t)) return GFSHIP_EINVAL;
  /* tracers: van Leer gradient unless the file says otherwise + gfs_face_advection_flux
     (src/variable.c:427-431) */
  int gradient = 1;
  for (size_t q = 0; q < s->tracers.size (); q++)
    if (s->tracers[q] == t) gradient = s->tracer_gradient[q];
  TRY (variable_sources (s, t, t, gradient, false, dt, nullptr, nullptr));
  TRY (bc_leaf (s, t));
  return GFSHIP_OK;
}

static int domain_cfl (gfship_sim * s, double * cfl, bool cached)
{
  double * u[3], * un[3];
  ptrs3 (s, s->u, u);
  ptrs3 (s, s->un, un);
  double c2;
  if (cached && s->cfl_ready)
    TRY (launch_cfl_from_max (s->dom, &c2));
  else {
    TRY (materialize_un (s));
    TRY (launch_cfl (s->dom, u, un, s->visc, &c2));
  }
  s->cfl_ready = false;
  *cfl = sqrt (c2);
  return GFSHIP_OK;
}

int gfship_domain_cfl (gfship_sim * s, double * cfl)
{
  GFSHIP_CHECK (s && cfl, GFSHIP_EINVAL, "null argument");
  double * u[3], * un[3];
  ptrs3 (s, s->u, u);
  ptrs3 (s, s->un, un);
  double c2;
  s->cfl_ready = false;
  TRY (materialize_un (s));
  TRY (launch_cfl (s->dom, u, un, s->visc, &c2));
  *cfl = sqrt (c2);
  return GFSHIP_OK;
}

// cached: the velocities have not changed since the approximate projection that has just run
// (only gfship_sim_start / gfship_sim_step say so), its maxima give the CFL time scale
static int set_timestep (gfship_sim * s, bool cached)
{
  double t = s->t;
  double cfl = s->advection_params.cfl;
  if (cfl < DBL_MAX) {
    double dcfl;
    TRY (domain_cfl (s, &dcfl, cached));
    s->advection_params.dt = cfl*dcfl;
  }
  else
    s->advection_params.dt = G_MAXINT;
  if (s->advection_params.dt > s->dtmax)
    s->advection_params.dt = s->dtmax;

  /* the events bound the step (gfs_event_next loop, src/simulation.c:1603-1610): the host that
     owns the GfsEvent list answers through the hook */
  double tnext = G_MAXINT;
  if (s->next_event)
    tnext = s->next_event (s->next_event_ctx, t, s->i);
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
  return GFSHIP_OK;
}

int gfship_set_timestep (gfship_sim * s)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  return set_timestep (s, false);
}

static void sim_variables (gfship_sim * s, std::vector<Field *> & v)
{
  gfship_domain * dom = s->dom;
  v.push_back (get_field (dom, s->p));
  v.push_back (get_field (dom, s->pmac));
  for (int c = 0; c < dom->dim; c++)
    v.push_back (get_field (dom, s->u[c]));
  for (gfship_field t : s->tracers)
    v.push_back (get_field (dom, t));
}

int gfship_sim_set_tracer_gradient (gfship_sim * s, int t, int gradient)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  GFSHIP_CHECK (t >= 0 && (size_t) t < s->tracers.size (), GFSHIP_EINVAL, "no tracer %d", t);
  GFSHIP_CHECK (gradient >= 0 && gradient <= 4, GFSHIP_EINVAL,
		"gradient: 0 centred, 1 van Leer, 2 minmod, 3 superbee, 4 sweby");
  s->tracer_gradient[t] = gradient;
  return GFSHIP_OK;
}

static int coarse_init (gfship_sim * s, bool u_done);

int gfship_coarse_init (gfship_sim * s)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  return coarse_init (s, false);
}

// u_done: the level below the leaves of U, V, W has just been filled (correct_centered_coarse_kernel)
static int coarse_init (gfship_sim * s, bool u_done)
{
  std::vector<Field *> v;
  sim_variables (s, v);
  /* Pmac is not touched between this point of the loop and the MAC projection of the next step:
     its non-leaf values, whenever somebody reads them (a snapshot, a download of a coarse level:
     coarse_flush), are the ones computed here -- so they are computed then, not in every step */
  Field * pmac = get_field (s->dom, s->pmac);
  std::vector<Field *> now, vel;
  for (Field * F : v) {
    bool is_u = false;
    for (int c = 0; c < s->dom->dim; c++)
      if (F == get_field (s->dom, s->u[c])) is_u = true;
    if (F == pmac) F->coarse_stale = true;
    else {
      F->coarse_stale = false;
      F->coarse_valid = true;
      if (u_done && is_u) vel.push_back (F); else now.push_back (F);
    }
  }
  if (vel.empty ())
    return launch_coarse_init (s->dom, now.data (), (int) now.size ());
  /* the level below the leaves for the others, then every variable together on the levels below */
  TRY (launch_coarse_init_levels (s->dom, now.data (), (int) now.size (), s->dom->depth - 1, s->dom->depth - 1));
  now.insert (now.end (), vel.begin (), vel.end ());
  if (s->dom->depth >= 2)
    TRY (launch_coarse_init_levels (s->dom, now.data (), (int) now.size (), s->dom->depth - 2, 0));
  return GFSHIP_OK;
}

int gfship_sim_start (gfship_sim * s)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  GFSHIP_CHECK (s->advection_params.gc, GFSHIP_EUNSUPPORTED, "gc = 0 is not supported");
  /* gfs_simulation_init (src/simulation.c:1068-1103): BCs on every variable, coarse cells */
  TRY (bc_leaf (s, s->p));
  TRY (bc_leaf (s, s->pmac));
  for (int c = 0; c < s->dom->dim; c++)
    TRY (bc_leaf (s, s->u[c]));
  for (gfship_field t : s->tracers)
    TRY (bc_leaf (s, t));
  /* gfs_cell_coarse_init at this point of the loop (src/simulation.c:530-533): the non-leaf
     values are those of the state before the approximate projection, so it cannot be deferred */
  TRY (gfship_coarse_init (s));

  TRY (gfship_set_timestep (s));
  if (s->i == 0) {
    TRY (gfship_approximate_projection (s, &s->approx_projection_params, s->advection_params.dt,
					s->p, s->g));
    TRY (set_timestep (s, true));
    TRY (advance_tracers (s, s->advection_params.dt/2.));
  }
  else {
    /* a simulation read back from a snapshot (time.i > 0): gfs_update_gradients
       (src/simulation.c:474-475, src/timestep.c:306-322) rebuilds the centred pressure gradient g
       from P -- gfs_correct_normal_velocities with dt = 0 leaves the face velocities alone and
       accumulates the same face differences as the approximate projection that ended the step the
       snapshot was taken after, then gfs_scale_gradients -- so the restarted run continues bit for
       bit */
    double * gp[3];
    ptrs3 (s, s->g, gp);
    if (s->has_alpha) {
      TRY (gfship_poisson_coefficients_alpha (s->dom, s->alpha));
      TRY (launch_project_correct_weighted (s->dom, leaf (s, s->p), nullptr, gp, nullptr, 0.));
    }
    else {
      TRY (gfship_poisson_coefficients (s->dom));
      TRY (launch_centered_gradient (s->dom, leaf (s, s->p), gp));
    }
    for (int c = 0; c < s->dom->dim; c++)
      TRY (bc_leaf (s, s->g[c]));
  }
  return GFSHIP_OK;
}

int gfship_sim_restart (gfship_sim * s, double t, unsigned i)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  s->t = s->tnext = t;
  s->i = i;
  return GFSHIP_OK;
}

// loop body of advection_run (src/simulation.c:2078-2111) of a GfsAdvection simulation whose MAC
// velocities are given (GfsVariableStreamFunction: uploaded through GFSHIP_VAR_UN and kept as they
// are): coarse values, time step, tracers
int gfship_sim_advection_step (gfship_sim * s)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  TRY (materialize_un (s));
  TRY (gfship_coarse_init (s));
  TRY (set_timestep (s, false));
  TRY (advance_tracers (s, s->advection_params.dt));
  s->t = s->tnext;
  s->i++;
  return GFSHIP_OK;
}

int gfship_sim_step (gfship_sim * s)
{
  GFSHIP_CHECK (s != nullptr, GFSHIP_EINVAL, "null simulation");
  gfship_domain * dom = s->dom;
  const gfship_field * gc = s->g;

  TRY (predicted_face_velocities (s, s->advection_params.dt/2.));

  /* gfs_variables_swap (p, pmac); gfs_mac_projection (...dt/2, p, gmac); swap back
     (src/simulation.c:498-503): the projection runs on Pmac's storage with P's BCs */
  {
    Field * P = get_field (dom, s->p), * PM = get_field (dom, s->pmac);
    std::swap (P->lev, PM->lev);
    int r = mac_projection (s, &s->projection_params, s->advection_params.dt/2., s->p, s->gmac);
    std::swap (P->lev, PM->lev);
    if (r != GFSHIP_OK) return r;
  }

  bool u_coarse = false, corrected = false;
  TRY (centered_velocity_advection (s, s->gmac, s->i > 0 ? gc : s->gmac, - s->advection_params.dt,
				    &corrected, &u_coarse));
  if (!corrected)
    TRY (correct_centered_velocities (s, s->i > 0 ? gc : s->gmac, - s->advection_params.dt, &u_coarse));

  /* gfs_cell_coarse_init at this point of the loop (src/simulation.c:530-533): the non-leaf
     values are those of the state before the approximate projection, so it cannot be deferred */
  TRY (coarse_init (s, u_coarse));

  TRY (mac_projection (s, &s->approx_projection_params, s->advection_params.dt, s->p, s->g, true, true));
  s->t = s->tnext;
  s->i++;

  TRY (set_timestep (s, true));
  TRY (advance_tracers (s, s->advection_params.dt));
  return GFSHIP_OK;
}

int gfship_divergence_norm (gfship_sim * s, gfship_norm * out)
{
  GFSHIP_CHECK (s && out, GFSHIP_EINVAL, "null argument");
  double * u[3];
  ptrs3 (s, s->u, u);
  TRY (launch_velocity_divergence (s->dom, u, leaf (s, s->div)));
  return gfship_norm_variable (s->dom, s->div, out);
}

int gfship_sim_download_un (gfship_sim * s, int c, double * host)
{
  GFSHIP_CHECK (s && host, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (c >= 0 && c < s->dom->dim, GFSHIP_EINVAL, "component %d out of range", c);
  TRY (materialize_un (s));
  return gfship_field_download (s->dom, s->un[c], s->dom->depth, host);
}

} // extern "C"
