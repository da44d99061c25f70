/* go_sim.h -- oracle: state of one GfsSimulation on a uniform box (simulation.h:56-82).
 * TEST INFRASTRUCTURE ONLY (see gfs_oracle.h). */
#ifndef GO_SIM_H
#define GO_SIM_H
#include "gfs_oracle.h"

#define GO_MAXTRACERS 4

typedef struct GoSim {
  GoDomain * dom;
  GoField * p, * pmac, * u[3], * g[3], * gmac[3];
  GoField * tracer[GO_MAXTRACERS];
  int ntracers;
  int tracer_gradient[GO_MAXTRACERS];   /* GfsVariableTracer { gradient = }: 0 centred, 1 van Leer (default) */
  double * un[6];   /* GFS_STATE (cell)->f[d].un on leaves, ghosts included */
  double * fv[6];   /* GFS_STATE (cell)->f[d].v  on leaves, ghosts included */
  GoMultilevelParams projection_params, approx_projection_params;
  GoAdvectionParams advection_params;
  double t, end, dtmax, tnext;   /* GfsTime */
  double visc[3];                /* SourceDiffusion {} U nu: constant diffusion coefficient, 0 = none */
  double src[3];                 /* GfsSource {} U g: constant intensity (src/source.c:398-403,476-481), 0 = none */
  GoMultilevelParams diffusion_params[3]; /* GfsDiffusion.par (source.c:966-974) */
  unsigned i, iend;
  /* GfsPhysicalParams { alpha = ... } (simulation.c:1306-1440) as gfs_function_face_value (alpha) on the
   * leaf faces: alpha[c] in the layout go_poisson_coefficients_alpha takes; NULL = alpha NULL */
  GoField * alpha[3];
} GoSim;

GoSim * go_sim_new (int dim, int depth, const int side[6]);
void    go_sim_destroy (GoSim * s);
GoField * go_sim_field (GoSim * s, int which, int c);
GoDomain * go_sim_domain (GoSim * s);
GoMultilevelParams * go_sim_projection_params (GoSim * s);
GoMultilevelParams * go_sim_approx_projection_params (GoSim * s);
GoAdvectionParams * go_sim_advection_params (GoSim * s);
double * go_sim_un (GoSim * s, int d);
double * go_sim_fv (GoSim * s, int d);
double  go_sim_time (GoSim * s);
unsigned go_sim_iter (GoSim * s);
void    go_sim_set_time (GoSim * s, double end, double dtmax);
int     go_sim_add_tracer (GoSim * s);
void    go_sim_set_tracer_gradient (GoSim * s, int t, int gradient);
void    go_approximate_projection (GoSim * s, GoMultilevelParams * par, double dt, GoField * p,
				   GoField ** g);
void    go_mac_projection (GoSim * s, GoMultilevelParams * par, double dt, GoField * p,
			   GoField ** g);
void    go_predicted_face_velocities (GoSim * s);
void    go_centered_velocity_advection (GoSim * s, GoField ** gmac, GoField ** g);
void    go_tracer_advection (GoSim * s, GoField * t, double dt);
double  go_domain_cfl (GoSim * s);
void    go_set_timestep (GoSim * s);
void    go_coarse_init (GoSim * s);
void    go_sim_start (GoSim * s);
void    go_sim_restart (GoSim * s, double t, unsigned i);
void    go_sim_step (GoSim * s);
void    go_advection_step (GoSim * s);
void    go_divergence (GoSim * s, GoField * out);
void    go_sim_set_viscosity (GoSim * s, int c, double nu);
void    go_sim_set_source (GoSim * s, int c, double g);
void    go_sim_set_alpha (GoSim * s, GoField * const alpha[3]);
GoMultilevelParams * go_sim_diffusion_params (GoSim * s, int c);
/* go_diffusion.c */
double  go_source_diffusion_value (GoSim * s, GoField * phi, int cell, double D);
void    go_variable_diffusion (GoSim * s, GoField * v, GoField * rhs, double D, double dt,
			       GoMultilevelParams * par);
#endif
