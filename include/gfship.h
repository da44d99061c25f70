/* gfship.h -- C ABI of libgfship: the MI355X (gfx950) projection / advection / particle path
 * of Gerris behind the reference's own solver-plugin boundary.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference tree).  Plain C types only: opaque domain handle, integer field handles, host
 * pointers and sizes.  All functions returning int return 0 on success and a negative
 * GFSHIP_E* code on failure; gfship_last_error() then holds a message (the reference's hooks
 * return void and report through g_warning / g_log).
 *
 * The library is device-only: there is no CPU fallback.  gfship_domain_create() fails with
 * GFSHIP_ENODEVICE when no HIP device is present.
 *
 * Host-side array convention (upload/download): level l of a field is (n+2)^dim doubles,
 * n = 2^l, x fastest, one ghost layer per side; cell (i,j,k), 1 <= i,j,k <= n, is at
 * i + (n+2)*(j + (n+2)*k).  j grows with y, k with z.  The box is the unit cube/square
 * centred on the origin (GfsBox of size 1, boundary.h:319-327).
 */
#ifndef GFSHIP_H
#define GFSHIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFSHIP_VERSION   1
#define GFSHIP_MAXLEVEL  12

enum { GFSHIP_OK = 0, GFSHIP_EINVAL = -1, GFSHIP_ENODEVICE = -2, GFSHIP_ENOMEM = -3,
       GFSHIP_EHIP = -4, GFSHIP_EUNSUPPORTED = -5 };

/* what is behind each side of the box: box->neighbor[d] (src/boundary.h:319-327) */
enum { GFSHIP_SIDE_PERIODIC = 0,   /* GfsBoundaryPeriodic self edge, src/boundary.c:1704-1760 */
       GFSHIP_SIDE_BOUNDARY = 1,   /* GfsBoundary + per-variable GfsBc (default symmetry)    */
       GFSHIP_SIDE_EXTERNAL = 2 }; /* GfsBoundaryMpi, src/mpi_boundary.c:78-246              */

/* GfsBc kinds on a GFSHIP_SIDE_BOUNDARY side: src/boundary.c:45-74 (symmetry, the default),
   :253-279 (GfsBcDirichlet), :336-360 (GfsBcNeumann) */
enum { GFSHIP_BC_SYMMETRY = 0, GFSHIP_BC_DIRICHLET = 1, GFSHIP_BC_NEUMANN = 2 };

/* smoother of gfship_relax / gfship_poisson_cycle:
   EXACT  reproduces the reference's in-place sweep (src/poisson.c:507-557, tree pre-order of
          src/ftt.c:837-852) bit for bit on the device;
   REDBLACK is a two-colour Gauss-Seidel of the same operator: same fixed point, different
          iterates (not a reference algorithm; opt-in);
   EXACT_HYPERPLANE is EXACT with one launch per hyperplane instead of the pipelined
          tile sweep (same bits; kept as an independent implementation to test against);
   EXACT_PER_SWEEP is EXACT with one launch per sweep of a relax loop even where the sweeps of
          the loop can be pipelined in a single launch, and with the advection step through the
          face-value arrays even where it can be fused (same bits; the general paths, kept
          selectable to test the fused ones against them). */
enum { GFSHIP_RELAX_EXACT = 0, GFSHIP_RELAX_REDBLACK = 1, GFSHIP_RELAX_EXACT_HYPERPLANE = 2,
       GFSHIP_RELAX_EXACT_PER_SWEEP = 3 };

typedef struct gfship_domain gfship_domain;   /* GfsDomain + its per-level SoA device arrays */
typedef int gfship_field;                     /* GfsVariable handle (>= 0) */

typedef struct {           /* GfsNorm, src/fluid.h; filled as src/fluid.c:2107-2171 */
  double bias, first, second, infty, w;
} gfship_norm;

typedef struct {           /* GfsMultilevelParams, src/poisson.h:39-52, field for field */
  double tolerance;
  unsigned nrelax, erelax;
  unsigned minlevel;
  unsigned nitermax, nitermin;
  unsigned dimension;
  unsigned niter;
  unsigned depth;
  int weighted, function;
  double beta, omega;
  gfship_norm residual_before, residual;
} gfship_multilevel_params;

const char * gfship_last_error (void);
int          gfship_version (void);
int          gfship_device_count (void);

/* ---- domain and variables -------------------------------------------------------------- */

/* One uniform GfsBox refined to `depth` (Refine depth): replaces the FttCell/FttOct tree of
   src/ftt.h:134-159 and the per-cell GfsStateVector of src/fluid.h:39-52 by per-level SoA
   device arrays.  side[d], d = 0..2*dim-1 in the order of src/ftt.h:78-89. */
int  gfship_domain_create (gfship_domain ** dom, int dim, int depth, const int side[6],
			   int device);
void gfship_domain_destroy (gfship_domain * dom);
int  gfship_domain_set_relax_mode (gfship_domain * dom, int mode);
int  gfship_domain_synchronize (gfship_domain * dom);
/* raw HIP stream (hipStream_t) all work of this domain is enqueued on */
void * gfship_domain_stream (gfship_domain * dom);

/* gfs_domain_add_variable / gfs_temporary_variable (src/domain.c:3271-3385);
   component = 0..2 for a vector component (gfs_variable_set_vector), -1 for scalars */
gfship_field gfship_field_alloc (gfship_domain * dom, int component);
int  gfship_field_free (gfship_domain * dom, gfship_field f);
/* GfsBc of variable f on side d (src/boundary.c:1853-2014 box reader); val = n^(dim-1)
   leaf-level face-centre values (first tangential axis fastest) or NULL for 0 */
int  gfship_field_set_bc (gfship_domain * dom, gfship_field f, int d, int type,
			  const double * val);
int  gfship_field_upload (gfship_domain * dom, gfship_field f, int level, const double * host);
int  gfship_field_download (gfship_domain * dom, gfship_field f, int level, double * host);
int  gfship_field_fill (gfship_domain * dom, gfship_field f, int level, double value);
/* device pointer and layout of level `level`: cell (i,j,k) is at
   ptr[xo + i + px*(j + (n+2)*k)] */
void * gfship_field_device_ptr (gfship_domain * dom, gfship_field f, int level,
				int * px, int * xo);

/* gfs_domain_copy_bc (src/domain.c:846-867) / gfs_domain_homogeneous_bc (:945-965) */
int  gfship_bc (gfship_domain * dom, gfship_field v, gfship_field v1, int level);
int  gfship_homogeneous_bc (gfship_domain * dom, gfship_field ov, gfship_field v, int level);

/* ---- Poisson multigrid (the GfsPoissonSolverFunc boundary, src/poisson.h:32-38) -------- */

void gfship_multilevel_params_init (gfship_multilevel_params * par, int dim);
						/* gfs_multilevel_params_init, src/poisson.c:70-89 */
/* gfs_poisson_coefficients (src/poisson.c:856-901) with alpha = NULL */
int  gfship_poisson_coefficients (gfship_domain * dom);
/* gfs_poisson_coefficients (src/poisson.c:856-901) with a GfsFunction alpha (the inverse of the
   density: the equation is div (alpha grad p) = ...): alpha[c] holds gfs_function_face_value (alpha,
   face) of the leaf faces normal to c in the layout of a variable (the entry of a cell is its + face
   along c, the ghost entry in front of the first cell its - face).  Sets f[d].v of every cell of
   every level (poisson_coeff :769-797, face_coeff_from_below :826-853 with its one-neighbour rule);
   relax, residual, cycle and solve then use them (exact sweep order; 3-D levels of 32^3 and more on the
   pipelined tile kernel with the six weight rows streamed beside u / rhs / dia, 2-D levels in one
   launch per sweep).  alpha = NULL: gfship_poisson_coefficients.  The projections of a gfship_sim take
   the alpha of gfship_sim_set_alpha.  gfship_poisson_weights: the variable holding f[d].v,
   d = 0 .. 2 dim - 1. */
int  gfship_poisson_coefficients_alpha (gfship_domain * dom, const gfship_field alpha[3]);
int  gfship_poisson_weights (gfship_domain * dom, int d, gfship_field * w);
/* gfs_relax, src/poisson.c:604-632: one in-place sweep of level `level` */
int  gfship_relax (gfship_domain * dom, unsigned d, int level, double omega,
		   gfship_field u, gfship_field rhs, gfship_field dia);
/* gfs_residual, src/poisson.c:721-747 */
int  gfship_residual (gfship_domain * dom, unsigned d, int level,
		      gfship_field u, gfship_field rhs, gfship_field dia, gfship_field res);
/* gfs_domain_norm_residual, src/domain.c:2239-2288 */
int  gfship_norm_residual (gfship_domain * dom, double dt, gfship_field res, gfship_norm * out);
/* gfs_domain_norm_variable on leaves, src/domain.c:2197-2232 */
int  gfship_norm_variable (gfship_domain * dom, gfship_field v, gfship_norm * out);
/* Two properties of the pipelined sweeps behind gfship_relax / _cycle / _solve on 3-D levels of 32^3
   cells and more, stated here because a caller can run into them:
   - their tiles hand lines over through 8-byte words in which the value itself is the flag: a word
     of all ones (0xFFFFFFFFFFFFFFFF, a NaN with that particular payload, which no arithmetic of the
     solver produces) means "not there yet".  A field that holds exactly that bit pattern in a cell
     along a tile border makes the hand-off wait run into its bound: the call returns GFSHIP_EHIP with
     the level in gfship_last_error().  Ordinary NaNs and infinities pass through like any value;
   - the loop of a periodic level in one launch needs all its tiles (256 workgroups at 256^3) resident
     on the device at the same time.  The library checks the occupancy and makes one trial run per
     level; if another process or stream holds compute units later on, the bounded waits time out,
     that solve fails with GFSHIP_EHIP (the field may then hold partially updated values) and the
     domain falls back to one launch per sweep for the rest of its life -- results of later calls
     are unaffected (same arithmetic, same order). */
/* gfs_poisson_cycle, src/poisson.c:1109-1178 */
int  gfship_poisson_cycle (gfship_domain * dom, gfship_multilevel_params * p,
			   gfship_field u, gfship_field rhs, gfship_field dia, gfship_field res);
/* gfs_poisson_solve, src/poisson.c:1225-1269: the function installed in
   GfsMultilevelParams.poisson_solve (called at src/timestep.c:423, src/simulation.c:2267) */
int  gfship_poisson_solve (gfship_domain * dom, gfship_multilevel_params * par,
			   gfship_field lhs, gfship_field rhs, gfship_field res,
			   gfship_field dia, double dt);

/* ---- implicit diffusion (GfsSourceDiffusion with a constant coefficient, no solids) ---------- */

/* gfs_diffusion_coefficients, src/poisson.c:1357-1390: face weights lambda2*beta*dt*D on every
   level (one scalar per level on a uniform box) and rhoc = 1. on every cell */
int  gfship_diffusion_coefficients (gfship_domain * dom, double D, double dt, gfship_field rhoc,
				    double beta);
/* gfs_diffusion_rhs, src/poisson.c:1447-1453 (diffusion_rhs :1392-1436) */
int  gfship_diffusion_rhs (gfship_domain * dom, gfship_field v, gfship_field rhs,
			   gfship_field rhoc, double beta);
/* gfs_diffusion_residual, src/poisson.c:1587-1612 (diffusion_residual :1519-1556) */
int  gfship_diffusion_residual (gfship_domain * dom, gfship_field u, gfship_field rhs,
				gfship_field rhoc, gfship_field res);
/* gfs_diffusion_cycle, src/poisson.c:1633-1690 */
int  gfship_diffusion_cycle (gfship_domain * dom, unsigned levelmin, unsigned depth,
			     unsigned nrelax, gfship_field u, gfship_field rhs, gfship_field rhoc,
			     gfship_field res);
/* gfs_diffusion, src/timestep.c:735-788: the function installed in
   GfsAdvectionParams.diffusion_solve (called at src/timestep.c:944) */
int  gfship_diffusion (gfship_domain * dom, gfship_multilevel_params * par, gfship_field v,
		       gfship_field rhs, gfship_field rhoc);

/* ---- projection + advection time step (src/timestep.c, src/advection.c, src/simulation.c) -- */

typedef struct gfship_sim gfship_sim;   /* GfsSimulation on one box, src/simulation.h:56-82 */

typedef struct {           /* the fields of GfsAdvectionParams used here, src/advection.h:50-69 */
  double cfl, dt;
  int gradient;            /* 0 gfs_center_gradient (src/fluid.c:434-475),
			      1 gfs_center_van_leer_gradient (:522-561), 2 gfs_center_minmod_gradient,
			      3 gfs_center_superbee_gradient, 4 gfs_center_sweby_gradient (:563-690) */
  int gc;
} gfship_advection_params;

enum { GFSHIP_VAR_P = 0, GFSHIP_VAR_PMAC = 1, GFSHIP_VAR_U = 2, GFSHIP_VAR_G = 3,
       GFSHIP_VAR_GMAC = 4, GFSHIP_VAR_TRACER = 5,
       GFSHIP_VAR_UN = 6 /* MAC velocity of the + face of every cell along c (ghost cell 0: the - face of
			    the first cell), GFS_STATE (cell)->f[2c].un */ };

/* gfs_simulation_new + simulation_init (src/simulation.c:910-1015): allocates P, Pmac, U, V(, W),
   the gradient vectors g[], gmac[] of simulation_run (:432-456) and the face state */
int  gfship_sim_create (gfship_sim ** sim, gfship_domain * dom);
void gfship_sim_destroy (gfship_sim * sim);
gfship_field gfship_sim_variable (gfship_sim * sim, int which, int c);
gfship_multilevel_params * gfship_sim_projection_params (gfship_sim * sim);
gfship_multilevel_params * gfship_sim_approx_projection_params (gfship_sim * sim);
gfship_advection_params *  gfship_sim_advection_params (gfship_sim * sim);
int      gfship_sim_set_time (gfship_sim * sim, double end, double dtmax); /* GfsTime */
/* gfs_simulation_set_timestep shortens the step so that it lands on the next event
   (src/simulation.c:1603-1610: for every GfsEvent, `if (t < next && next < tnext) tnext = next +
   1e-9` with next = gfs_event_next, starting from G_MAXINT).  The events belong to the host:
   the hook returns that tnext for the simulation time t and iteration i it is given. */
typedef double (* gfship_next_event_fn) (void * ctx, double t, unsigned i);
int      gfship_sim_set_next_event (gfship_sim * sim, gfship_next_event_fn fn, void * ctx);
double   gfship_sim_time (gfship_sim * sim);
unsigned gfship_sim_iter (gfship_sim * sim);
int      gfship_sim_add_tracer (gfship_sim * sim);       /* GfsVariableTracer, src/variable.c:427-431 */
/* GfsVariableTracer { gradient = gfs_center_gradient | gfs_center_van_leer_gradient (default) |
   gfs_center_minmod_gradient | gfs_center_superbee_gradient | gfs_center_sweby_gradient }: 0 .. 4
   (src/fluid.c:434-690); the same numbers in gfship_advection_params.gradient */
int      gfship_sim_set_tracer_gradient (gfship_sim * sim, int tracer, int gradient);
/* GfsSourceDiffusion {} U|V|W nu (src/source.c:933-1160): constant implicit viscosity of
   velocity component c (0. removes it), and the GfsMultilevelParams of its solver
   (tolerance 1e-6, beta 1: diffusion_init, src/source.c:966-974) */
int      gfship_sim_set_viscosity (gfship_sim * sim, int c, double nu);
/* GfsPhysicalParams { alpha = ... } (src/simulation.c:1306-1440): the inverse of the density as
   gfs_function_face_value (alpha, face) on the leaf faces, alpha[c] in the layout of
   gfship_poisson_coefficients_alpha (kept by handle: the caller may rewrite the fields between steps
   when alpha depends on time or on a tracer).  Both projections then run gfs_poisson_coefficients
   with it (src/timestep.c:376), the multigrid with the face weights, and gfs_correct_normal_velocities
   / gfs_update_gradients with gfs_face_weighted_gradient's weights (:118-144,306-322).  NULL: alpha =
   NULL again.  Not together with GfsSourceDiffusion (variable-density diffusion coefficients). */
int      gfship_sim_set_alpha (gfship_sim * sim, const gfship_field alpha[3]);
/* GfsSource {} U|V|W g (src/source.c:362-500) with a constant intensity g on velocity component c
   (0. removes it): a body force per unit mass -- the MAC source of gfs_cell_advected_face_values
   (src/advection.c:88, gfs_variable_mac_source), the centred source added at the end of
   variable_sources (gfs_domain_variable_centered_sources, src/source.c:62-108), the acceleration
   time scale of gfs_domain_cfl (src/domain.c:2893-2901), and the gravity GfsForceBuoy reads
   (modules/particulatecommon.c:632-647).  Pinned by test/poiseuille (error.ref). */
int      gfship_sim_set_source (gfship_sim * sim, int c, double intensity);
gfship_multilevel_params * gfship_sim_diffusion_params (gfship_sim * sim, int c);
/* simulation_run before its loop (src/simulation.c:458-476): BCs, first time step, initial
   approximate projection */
int  gfship_sim_start (gfship_sim * sim);
/* GfsTime { i = .. t = .. } of a simulation file that is a snapshot of a running simulation
   (gfs_time_read, src/simulation.c:1687-1725): call it before gfship_sim_start, which then takes
   simulation_run's `time.i > 0' branch (no initial projection; gfs_update_gradients,
   src/simulation.c:474-475) */
int  gfship_sim_restart (gfship_sim * sim, double t, unsigned i);
/* one iteration of the simulation_run loop body (src/simulation.c:479-548) */
int  gfship_sim_step (gfship_sim * sim);
/* loop body of advection_run (GfsAdvection, src/simulation.c:2078-2111) with given MAC velocities
   (GfsVariableStreamFunction, src/variable.c:931-1086: upload them through GFSHIP_VAR_UN and the
   centred ones through GFSHIP_VAR_U): coarse values, time step, tracers */
int  gfship_sim_advection_step (gfship_sim * sim);
/* the pieces, callable on their own: */
int  gfship_predicted_face_velocities (gfship_sim * sim);      /* src/timestep.c:681-717 */
int  gfship_mac_projection (gfship_sim * sim, gfship_multilevel_params * par, double dt,
			    gfship_field p, const gfship_field g[3]);        /* :460-484 */
int  gfship_approximate_projection (gfship_sim * sim, gfship_multilevel_params * par, double dt,
				    gfship_field p, const gfship_field g[3]); /* :560-596 */
int  gfship_centered_velocity_advection (gfship_sim * sim, const gfship_field gmac[3],
					 const gfship_field g[3]);           /* :976-1016 */
int  gfship_tracer_advection (gfship_sim * sim, gfship_field t, double dt);  /* :1028-1055 */
int  gfship_domain_cfl (gfship_sim * sim, double * cfl);       /* src/domain.c:2824-2923 */
int  gfship_set_timestep (gfship_sim * sim);                   /* src/simulation.c:1569-1633 */
int  gfship_coarse_init (gfship_sim * sim);                    /* src/adaptive.c:43-58 */
/* OutputScalarNorm { v = Divergence }: gfs_divergence (src/fluid.c:2357-2376) + norm */
int  gfship_divergence_norm (gfship_sim * sim, gfship_norm * out);
/* MAC normal velocity on the faces orthogonal to component c, as a host array in the cell
   convention: entry (i,j,k) is the face on the + side of cell (i,j,k) (0 <= i <= n along c) */
int  gfship_sim_download_un (gfship_sim * sim, int c, double * host);

/* GfsOutputLocation's sampling (src/output.c:1182-1199): gfs_domain_locate + gfs_interpolate
   (src/fluid.c:2983-3101) of variable v at np points (pos = 3*np doubles); inside[q] = 0 where
   the point is outside the domain (out[q] is then 0) */
int  gfship_field_interpolate (gfship_domain * dom, gfship_field v, int np, const double * pos,
			       double * out, unsigned char * inside);

/* ---- snapshots: the cell data of a binary simulation file (GfsOutputSimulation { binary = 1 }) ---- */

/* What gfs_box_write / gfs_box_read put between the braces of the GfsBox when the domain parameter
   `binary = 1' is set (src/boundary.c:1819-1851,1853-2014): ftt_cell_write_binary
   (src/ftt.c:1771-1799) -- the tree in pre-order, children 0..7 in the order of src/ftt.c:301-316, per
   cell `guint flags' = child id | FTT_FLAG_LEAF on the deepest level -- with gfs_cell_write_binary
   (src/domain.c:3176-3207) as the per-cell function: a double -1. (no solid fractions), then one
   double per variable of `variables = ...'.  Every level of every variable is written / read (the
   non-leaf values are whatever gfs_cell_coarse_init left); ghost cells are not part of the file
   (apply the BCs after reading, as gfs_simulation_init does).
     gfship_snapshot_tree_bytes: size of that byte image for nvars variables;
     gfship_snapshot_tree_write: builds it on the device and copies it into host_buf;
     gfship_snapshot_tree_read: takes it apart into the variables; fails if the child ids, the leaf
       flags (the tree must be the uniform tree of this domain) or the -1. markers do not match
       (cell_read_binary src/ftt.c:1915-1945, gfs_cell_read_binary src/domain.c:3219-3270). */
size_t gfship_snapshot_tree_bytes (gfship_domain * dom, int nvars);
int  gfship_snapshot_tree_write (gfship_domain * dom, int nvars, const gfship_field * vars,
				 void * host_buf, size_t bytes);
int  gfship_snapshot_tree_read (gfship_domain * dom, int nvars, const gfship_field * vars,
				const void * host_buf, size_t bytes);

/* ---- energy spectra (modules/fft.c) ------------------------------------------------------------ */

/* GfsOutputEnergySpectra (modules/fft.c:1340-1474) of the whole box at the finest level: for each of
   the ncomp variables (U, V[, W]) the real-to-complex DFT of (u - <u>)/ntot (fill_cartesian_matrix,
   :966-1001), |F|^2 summed into Ek[knx^2 + kny^2 + kz^2] with the weights of :1409-1448.  Ek has
   gfship_energy_spectra_bins() entries; the reference prints "deltak*sqrt(i) Ek[i]" for i >= 1 and
   "# Total energy = Etot" (write_energy_spectra, :1340-1348).  The DFT is rocFFT's through hipFFT
   (FFTW in the reference): same definition, not the same rounding. */
int  gfship_energy_spectra_bins (gfship_domain * dom);
int  gfship_energy_spectra (gfship_domain * dom, int ncomp, const gfship_field * comps, double * Ek,
			    double * Etot, double * deltak);

/* GfsOutputSpectra (modules/fft.c:1101-1160) of the whole 3-D domain at the finest level: the r2c DFT
   (FFTW's sign and normalisation: none) of (v - <v>)/ntot, with <v> the volume average over the
   cells of all levels (substract_average, :897-908).  out receives N*N*(N/2 + 1) complex numbers
   (re, im), index (ix*N + iy)*(N/2 + 1) + iz, N = gfship_output_spectra_side(); *kstep = 2 pi/(x1 - x0)
   (init_kmax, :1031-1045): write_spectra prints "kx ky kz re*L im*L" with k = kstep times the
   signed index.  On a box of a lattice (communicator or gather hook) every rank receives the
   transform of the whole lattice -- the reference redistributes slabs for FFTW-MPI (:467-669) --
   and gfship_energy_spectra likewise bins the whole domain.
   gfship_output_spectra_plane: the same event with a flat box (realdim == 2, :1131-1141,
   fill_interpolated_cartesian_matrix :822-883): the cells the points of the plane normal to `normal' (0 x,
   1 y, 2 z) at coordinate pos lie in, minus their mean, over their number; the 2-D r2c DFT; out receives
   N*(N/2 + 1) complex numbers, index ia*(N/2 + 1) + ib, ia / ib the first / second in-plane coordinate
   (order_array keeps them in coordinate order behind the flat one); one box.  GfsOutputSpectraInterface
   samples the position of a VOF interface (out of scope with VOF): not provided. */
int  gfship_output_spectra_side (gfship_domain * dom);
int  gfship_output_spectra (gfship_domain * dom, gfship_field v, double * out, double * kstep);
int  gfship_output_spectra_plane (gfship_domain * dom, gfship_field v, int normal, double pos, double * out,
				  double * kstep);

/* GfsVariableTurbulentViscosity (modules/turbulence.c:953-1105): out = (Cs h)^2 |S| on the leaf cells
   from the centred differences of u (gfs_cm_gradient); model 1 = Smagorinsky (what the reference's
   files get: model_type is never read), model 0 = the sigma model of :980-1048 (libm calls: not
   bit-identical across C libraries). */
int  gfship_turbulent_viscosity (gfship_domain * dom, const gfship_field u[3], double Cs, int model,
				 gfship_field out);

/* GfsInitSpectra (modules/turbulence.c:270-901), 3-D: fills the variables v[0..2] with a synthetic
   solenoidal velocity field whose shell energies follow Pope's model spectrum (ReL != 0:
   alpha epsilon^(2/3) k^(-5/3) fL feta with c1, c2, c3) or k^2 (ReL = 0) below kmax, rescaled to the
   total energy E, on a periodic cube of side L centred on (x0, y0, z0) with 2^level points per side,
   interpolated at the cell centres.  The phases follow the reference: srand (seed) before every
   rand(), i.e. one value for all modes.  Field names in the struct = keywords of the .gfs object. */
typedef struct {
  double x0, y0, z0, L, E;
  double alpha, epsilon, c1, c2, c3, ReL, kmax, seed;
  int level;
} gfship_init_spectra_params;
int  gfship_init_spectra (gfship_domain * dom, const gfship_init_spectra_params * par,
			  const gfship_field v[3]);

/* ---- Lagrangian tracers (src/particle.c, modules/particulatecommon.c) ------------------------ */

typedef struct gfship_particles gfship_particles;   /* GfsParticleList of GfsParticle */

/* GfsParticleList read (modules/particulatecommon.c:1022-1093): np particles, pos = 3*np doubles
   (x y z per particle, z ignored in 2-D), id = np unsigned (src/particle.c:46-99 text format) */
int  gfship_particles_create (gfship_particles ** pl, gfship_sim * sim, int np,
			      const double * pos, const unsigned * id);
void gfship_particles_destroy (gfship_particles * pl);
/* gfs_particle_list_event (modules/particulatecommon.c:980-1015) without forces: drop particles
   outside the domain, RK2-advect each with the simulation's current velocity and dt
   (gfs_particle_event src/particle.c:31-44 -> gfs_domain_advect_point src/domain.c:2764-2788),
   then gfs_particle_bc (:3375-3395, periodic wrap :3189-3214) */
int  gfship_particle_list_event (gfship_particles * pl);
/* storage order only (no reference counterpart; arithmetic and download order unchanged): sort
   the slots by containing cell now / every `every` events (default 16, 0 = never) */
int  gfship_particles_sort (gfship_particles * pl);
int  gfship_particles_set_sort_interval (gfship_particles * pl, int every);
/* particles crossing a GFSHIP_SIDE_EXTERNAL side go to the box across it (send_particles /
   rcv_particles, modules/particulatecommon.c:3218-3312).  On a domain with the library's communicator
   (gfship_domain_comm_init) the packets travel by ncclSend / ncclRecv, counts first, then the records,
   and no hook is needed; without communicator and without hook the particles are dropped like at any
   non-periodic side.  A hook, when set, takes precedence.  It is called once per event by every box: nsend[d] records leave
   through side d (send[d]: gfship_particles_record_size doubles each -- position, old position, id,
   for particulates also velocity, mass, volume, force -- already in the
   coordinates of the receiving box, sorted by id); it returns in nrecv[d] / recv[d] the records
   that arrive through side d (host memory owned by the hook until its next call); they join the
   list in side order */
typedef int (* gfship_particle_migrate_fn) (void * ctx, const int nsend[6],
					    const double * const send[6], int nrecv[6],
					    const double * recv[6]);
int  gfship_particles_set_migrate (gfship_particles * pl, gfship_particle_migrate_fn fn, void * ctx);
/* doubles per record of the migration packets: 7, or 15 for particulates (+ velocity, mass, volume,
   force) */
int  gfship_particles_record_size (gfship_particles * pl);
/* slots in use (alive or not): upper bound of the count, size of the download buffers */
int  gfship_particles_slots (gfship_particles * pl);
int  gfship_particles_count (gfship_particles * pl);
/* positions and ids of the particles still on the list, in list order; returns their number */
int  gfship_particles_download (gfship_particles * pl, double * pos, unsigned * id);

/* GfsParticulate with forces (modules/particulatecommon.h:35-61, particulatecommon.c:91-842): the
   particles of the list get a velocity (3*np), a mass and a volume (np each), and the list a set of
   GfsParticleForce objects applied in the given order.  gfship_particle_list_event then runs
   gfs_particulate_event (:768-842): forces per unit volume from the fields at the start of the
   step -- GfsForceInertial Du/Dt (:285-336, with the velocity of the previous step kept in Un, Vn,
   Wn), GfsForceAddedMass (:363-427, cm = 0.5, adds rho*volume*cm to the mass at every event like the
   reference), GfsForceLift (:455-524, cl = 0.5), GfsForceDrag (:527-588, default law
   cd = 16 (1 + 0.15 Re^0.5)/Re below Re = 50, 48 (1 - 2.21/Re^0.5)/Re above; no force without
   viscosity), GfsForceBuoy (:619-653, gravity = the sum of the GfsSource intensities on U, V, W) --
   then pos += vel*dt/2, vel += force*dt/mass, pos += vel*dt/2 and gfs_particle_bc.  fluid density 1
   (alpha = NULL), viscosity = gfship_sim_set_viscosity of U.  Particulates migrate between boxes like
   tracers, with 15-double records.
   gfship_particles_set_force_coefficient: the GfsFunction a GfsForceAddedMass / GfsForceLift /
   GfsForceDrag object of the list may carry (gfs_force_coeff_read, :166-210) -- C text as in the
   simulation file, an expression or a { block } with a return, of the variables Rep, Urelp, Vrelp,
   Wrelp, Pdia (set per particle as in :364-384,462-485,545-573) and t.  gerris compiles it with the
   host compiler; here it is compiled for the GPU of the domain with hipRTC and evaluated for all
   particles by a kernel of its own before the event kernel (same C arithmetic, the device's libm).
   `force' is the index into the list given to gfship_particles_set_forces.  A text that does not
   compile: GFSHIP_EINVAL with the compiler's messages in gfship_last_error(). */
enum { GFSHIP_FORCE_INERTIAL = 1, GFSHIP_FORCE_ADDEDMASS = 2, GFSHIP_FORCE_LIFT = 3,
       GFSHIP_FORCE_DRAG = 4, GFSHIP_FORCE_BUOY = 5 };
int  gfship_particles_set_particulate (gfship_particles * pl, const double * vel, const double * mass,
				       const double * volume);
int  gfship_particles_set_forces (gfship_particles * pl, int nforces, const int * kinds,
				  const double gravity[3]);
int  gfship_particles_set_force_coefficient (gfship_particles * pl, int force, const char * function);
/* velocity (3 per particle), mass and force of the particles still on the list, in list order (any
   pointer may be NULL); returns their number */
int  gfship_particles_download_particulate (gfship_particles * pl, double * vel, double * mass,
					    double * force);

/* ---- one box per GPU: GfsBoundaryMpi sides (src/mpi_boundary.c:78-246) ----------------------- */

/* Either the in-library RCCL transport below (gfship_domain_comm_init: what bench.py uses), or two
   hooks of the caller for every GFSHIP_SIDE_EXTERNAL side (the tests implement them in-process and
   over gloo).  Semantics are the reference's parallel run with the
   domain parameter `overlap = 0` (src/domain.c:225,1104,1183): plain traversal order, ghost cells
   of MPI sides refreshed by every BC application, i.e. lagged by one sweep exactly like the
   periodic and MPI boundaries of the reference (send src/mpi_boundary.c:89-130, receive
   :132-222).
   exchange: fill the ghost layer of the level array `dev_ptr` (layout px, xo as in
     gfship_field_device_ptr) on external sides from the neighbour boxes' interior layer;
     kind = 0: a cell-centred variable, every external side (gfs_domain_copy_bc);
     kind = 1 + e: the face-value array fv[e] of gfs_domain_face_bc: only side e^1 is needed.
     Work must be ordered after everything already enqueued on gfship_domain_stream().
   reduce: MPI_Allreduce of vals[0..n-1] over all boxes, op = 0 sum, 1 max, 2 min
     (norms src/domain.c:2135-2166, CFL :2921). */
/* The domain parameter `overlap' of a parallel run (src/domain.c:225,682; "overlap = 0" in the
   graph parameters of the simulation file).  0 (the default here): every sweep of a relax loop in
   plain traversal order, BC application after it.  1 (the reference's default): the first
   nrelax - 1 sweeps of a relax loop in the order of gfs_traverse_and_homogeneous_bc
   (src/domain.c:1093-1125) -- the cells along the GfsBoundaryMpi sides first, side by side in
   traversal order, then the rest -- with the halo layers travelling beside the bulk of the sweep
   when the library's own communicator is used; the last sweep in plain order (src/poisson.c:1080-
   1086).  The two orders give different iterates of the same solution: a multi-box run must be
   compared with a reference run of the same setting.  No effect on a box without MPI sides. */
int  gfship_domain_set_overlap (gfship_domain * dom, int overlap);
typedef int (* gfship_exchange_fn) (void * ctx, void * dev_ptr, int level, int kind);
typedef int (* gfship_reduce_fn) (void * ctx, double * vals, int n, int op);
int  gfship_domain_set_exchange (gfship_domain * dom, gfship_exchange_fn fn, void * ctx);
int  gfship_domain_set_reduce (gfship_domain * dom, gfship_reduce_fn fn, void * ctx);
/* Optional third hook, together with the place of the box on the periodic lattice of boxes (rank r
   at (r % bx, (r / bx) % by, r / (bx by)), as for gfship_domain_comm_init below):
   gather: MPI_Allgather of `bytes' bytes of device memory per box: dev_recv + r*bytes receives what
     box r passed as dev_send; ordered after everything enqueued on gfship_domain_stream(), complete
     (or enqueued on that stream) on return.
   With it -- or with the library's own communicator -- the coarse end of a multigrid cycle (the
   levels of at most 16^3 cells per box: restrictions, relax loops with the BC application between
   their sweeps, prolongations; src/poisson.c:1131-1168) needs ONE collective instead of one halo
   exchange per sweep and level: every rank gathers the residuals of all boxes on the finest of
   those levels and runs the sweeps of every box of the lattice itself, handing the layers between
   the boxes over in device memory exactly where the reference's BC application would -- the same
   arithmetic in the same order on every rank, hence the same bits as one exchange per sweep.
   GFSHIP_NO_LATTICE_CYCLE=1 in the environment keeps one exchange per sweep. */
typedef int (* gfship_gather_fn) (void * ctx, const void * dev_send, void * dev_recv, size_t bytes);
int  gfship_domain_set_gather (gfship_domain * dom, gfship_gather_fn fn, void * ctx, int rank,
			       int nboxes, const int lattice[3]);
/* Which of the multi-box fast paths have run on this domain (for tests and benchmarks: both are
   bit-identical to the paths they replace): coarse ends of V-cycles computed for the whole lattice
   after one gather; launches of the tiled Godunov kernels with the face states beyond the MPI sides
   exchanged in one message per side (library communicator only, GFSHIP_NO_FUSED_MPI=1 disables). */
int  gfship_domain_path_counts (gfship_domain * dom, unsigned long long * lattice_cycles,
				unsigned long long * fused_mpi_launches);
/* The same boundary served inside the library over RCCL (xGMI on one node), no hooks: the domain
   is one GfsBox of a periodic lattice of lattice[0] x lattice[1] x lattice[2] boxes, one box per
   rank / GPU, rank r at (r % bx, (r / bx) % by, r / (bx by)) (gfs_domain_split, src/domain.c:2576-2599,
   one box per PE).  Every BC application becomes: pack kernel -> ncclSend / ncclRecv of the packed
   layers in one group (MPI_Isend / MPI_Recv of sndbuf / rcvbuf, src/mpi_boundary.c:89-222) -> unpack
   kernel, on the domain's stream; norms and the CFL minimum are reduced over the boxes
   (src/domain.c:2135-2166,2921) in one collective each.  Sides facing another box must be
   GFSHIP_SIDE_EXTERNAL; a communicator takes precedence over the hooks above.
     gfship_comm_available: opens RCCL (dlopen) without creating anything: GFSHIP_OK when
       gfship_domain_comm_init can be entered on this rank -- a launcher agrees on that over its own
       channel BEFORE the collective ncclCommInitRank, so that no rank waits in it alone;
     gfship_comm_unique_id: ncclGetUniqueId (GFSHIP_UNIQUE_ID_BYTES bytes): called by one rank, the
       bytes are then given to every rank (through the launcher's own channel: a file, a TCP store);
     gfship_domain_comm_init: ncclCommInitRank on the domain's device, collective over the ranks;
     gfship_domain_comm_size: ncclCommCount (0 without a communicator);
     gfship_domain_comm_stats: messages and bytes sent so far (domain->mpi_messages, mpi_size). */
#define GFSHIP_UNIQUE_ID_BYTES 128
int  gfship_comm_available (void);
int  gfship_comm_unique_id (void * id);
int  gfship_domain_comm_init (gfship_domain * dom, const void * unique_id, int rank, int nranks,
			      const int lattice[3]);
int  gfship_domain_comm_size (gfship_domain * dom);
int  gfship_domain_comm_stats (gfship_domain * dom, unsigned long long * messages,
			       unsigned long long * bytes);
int  gfship_domain_comm_destroy (gfship_domain * dom);

/* pack the interior layer adjacent to `side` of a level array into n^(dim-1) contiguous doubles
   (first tangential axis fastest) / unpack such a buffer into the ghost layer of `side`
   (the sndbuf / rcvbuf of src/boundary.c:1240-1258,1333-1347); device pointers, asynchronous on
   the domain's stream */
int  gfship_halo_pack (gfship_domain * dom, const void * dev_ptr, int level, int side, void * dev_buf);
int  gfship_halo_unpack (gfship_domain * dom, void * dev_ptr, int level, int side, const void * dev_buf);
/* the same for several sides with one kernel launch each way: sides[q] -> dev_bufs[q] */
int  gfship_halo_pack_sides (gfship_domain * dom, const void * dev_ptr, int level, int nsides,
			     const int * sides, void * const * dev_bufs);
int  gfship_halo_unpack_sides (gfship_domain * dom, void * dev_ptr, int level, int nsides,
			       const int * sides, void * const * dev_bufs);

/* ---- statically refined quadtree / octree (coarse-fine stencils) ---------------------------- */

/* A GfsSimulation on one periodic GfsBox whose tree is refined by a GfsRefine function instead of
   `Refine <int>' (the case of test/periodic/periodic.gfs with BOX = 1, 2; quadtree for dim = 2, octree
   for dim = 3): replaces the tree of FttCell / FttOct records (src/ftt.h:134-159) by one dense
   (n + 2)^dim array per level -- n = 2^l, one ghost layer, index i + (n + 2) (j + (n + 2) k) with
   1 <= i, j, k <= n inside, j growing with y and k with z (no k in 2-D) -- and a flag per
   cell: 0 no such cell, 1 leaf, 2 refined.  `refine' is the GfsFunction of the GfsRefine object:
   a cell whose level is below refine (x, y, z) at its centre is refined (refine_maxlevel,
   src/refine.c:35-38), with the constraints of the reference: neighbours differ by one level at most
   (oct_new, src/ftt.c:45-83) and the corner rule of ftt_refine_corner (src/ftt.c:2013-2074,
   src/simulation.c:1226-1231).  The algorithms are those of gfship_sim with the fine / coarse
   branches of src/fluid.c:64-93,178-197,283-309,364-396,778-893, src/advection.c:132-180,267-343,
   398-435,513-587, src/timestep.c:118-144 (in 3-D their FTT_3D forms: interpolate_2D1, src/fluid.c:214-245,
   src/advection.c:183-249): Euler equations, centred gradients, alpha = NULL, all sides periodic,
   refinement equal across each periodic pair (else GFSHIP_EUNSUPPORTED).  Results are those of the
   reference's traversal orders: the sweeps of gfs_relax run in tree order, the face loops accumulate
   in face-traversal order.  gfship_tree_host_check is not reentrant (it switches the library into a
   host-only mode while it runs); everything else follows the threading rule of the rest of the ABI:
   one thread per tree. */
typedef struct gfship_tree gfship_tree;
typedef double (* gfship_refine_fn) (double x, double y, double z, void * ctx);
enum { GFSHIP_TREE_P = 0, GFSHIP_TREE_PMAC, GFSHIP_TREE_U, GFSHIP_TREE_V, GFSHIP_TREE_GX, GFSHIP_TREE_GY,
       GFSHIP_TREE_GMACX, GFSHIP_TREE_GMACY, GFSHIP_TREE_UN0, GFSHIP_TREE_UN1, GFSHIP_TREE_UN2,
       GFSHIP_TREE_UN3, GFSHIP_TREE_W, GFSHIP_TREE_GZ, GFSHIP_TREE_GMACZ, GFSHIP_TREE_UN4,
       GFSHIP_TREE_UN5, GFSHIP_TREE_DIV, GFSHIP_TREE_BCVAL, GFSHIP_TREE_RES,
       GFSHIP_TREE_T0, GFSHIP_TREE_T1, GFSHIP_TREE_BCU, GFSHIP_TREE_BCV, GFSHIP_TREE_BCW };
				    /* variables of a tree: P, Pmac, U, V, g, gmac, f[d].un; then the 3-D ones;
				       DIV: the result of gfship_tree_divergence / the right-hand side of
				       gfship_tree_poisson_solve; BCVAL: the values of the conditions of P, one per
				       ghost cell (at the face centres); RES: the residual of the last solve;
				       T0, T1: the tracers of gfship_tree_add_tracer; BCU, BCV, BCW: the values of the
				       conditions of U, V, W, one per ghost cell (at the face centres) */
int  gfship_tree_create (gfship_tree ** tree, int dim, gfship_refine_fn refine, void * ctx, int device);
/* the same with GfsBoundary sides (side[d] = GFSHIP_SIDE_PERIODIC or GFSHIP_SIDE_BOUNDARY; the ghost
   cells of a boundary are refined like the cells they touch, gfs_domain_match): such a tree carries
   the Poisson problem of a GfsPoisson simulation (poisson_run, src/simulation.c:2213-2285; the case
   of test/poisson/circle/circle.gfs) -- gfship_tree_set_bc gives the condition of P on a side
   (GFSHIP_BC_SYMMETRY by default, _DIRICHLET, _NEUMANN: src/boundary.c:45-62,253-279,336-347), its
   values are uploaded as the variable GFSHIP_TREE_BCVAL, the right-hand side as GFSHIP_TREE_DIV, and
   gfship_tree_poisson_solve is gfs_poisson_solve (src/poisson.c:1225-1269) with dia = 0: the
   homogeneous conditions between the sweeps, the conditions themselves after each correction.
   gfship_tree_start / _step run on such a tree with the default conditions (GfsBc symmetry: slip
   walls -- the normal velocity component and its face values odd, everything else even,
   src/boundary.c:45-74); with a Dirichlet or Neumann condition set: GFSHIP_EUNSUPPORTED. */
int  gfship_tree_create_sides (gfship_tree ** tree, int dim, gfship_refine_fn refine, void * ctx,
			       const int * side, int device);
int  gfship_tree_set_bc (gfship_tree * tree, int d, int kind);
int  gfship_tree_poisson_solve (gfship_tree * tree, gfship_multilevel_params * par, double dt);
void gfship_tree_destroy (gfship_tree * tree);
int  gfship_tree_depth (const gfship_tree * tree);                   /* gfs_domain_depth */
int  gfship_tree_dim (const gfship_tree * tree);
int  gfship_tree_flags (const gfship_tree * tree, int level, unsigned char * out /* (n + 2)^dim */);
int  gfship_tree_upload (gfship_tree * tree, int var, int level, const double * in /* (n + 2)^dim */);
int  gfship_tree_download (gfship_tree * tree, int var, int level, double * out);
gfship_multilevel_params * gfship_tree_projection_params (gfship_tree * tree, int approx);
int  gfship_tree_set_time (gfship_tree * tree, double end, double cfl);  /* GfsTime end, AdvectionParams cfl */
int  gfship_tree_set_next_event (gfship_tree * tree, gfship_next_event_fn fn, void * ctx); /* as gfship_sim_set_next_event */
double   gfship_tree_time (const gfship_tree * tree);
double   gfship_tree_dt (const gfship_tree * tree);
unsigned gfship_tree_iter (const gfship_tree * tree);
/* simulation_run up to its loop (src/simulation.c:458-476) and one iteration of the loop (:479-548) */
/* GfsVariableTracer [{ gradient = }] on a tree (src/variable.c:427-431): advected with the MAC
   velocities at the end of every step and by half a step at the start (gfs_advance_tracers,
   src/simulation.c:405-430,476,548; gfs_tracer_advection_diffusion src/timestep.c:1028-1055 with
   gfs_face_advection_flux src/advection.c:356-381 -- flux/FTT_CELLS towards a coarse neighbour -- and the
   gradient 0 gfs_center_gradient / 1 gfs_center_van_leer_gradient (default) through gfs_neighbor_value,
   src/fluid.c:364-396,522-561), restricted by gfs_cell_coarse_init; default (symmetry) condition on
   GfsBoundary sides.  Returns GFSHIP_TREE_T0 / _T1 (two tracers at most), to be called before
   gfship_tree_start. */
int  gfship_tree_add_tracer (gfship_tree * tree, int gradient);
/* GfsBcDirichlet / GfsBcNeumann U|V|W on a GfsBoundary side of a tree (src/boundary.c:253-279,336-360 with
   their face_* forms for the Godunov face values; GFSHIP_BC_SYMMETRY: the default GfsBc), the values per
   ghost cell in GFSHIP_TREE_BCU + c; the conditions of P of gfship_tree_set_bc apply in the time step too
   (an outflow: BcDirichlet P 0 + BcNeumann U 0).
   GfsSourceDiffusion {} U|V|W nu on a tree: the implicit solve of gfs_diffusion (src/timestep.c:735-788,
   923-949, src/poisson.c:1271-1690: coefficients with their fine-coarse forms, gfs_diffusion_rhs,
   diffusion_relax in tree order with the homogeneous conditions of the component, gfs_diffusion_cycle with
   10 nrelax sweeps on the first level), the explicit term as MAC source of the face values and in the
   CFL condition.  Quadtrees only (GFSHIP_EUNSUPPORTED on an octree); with Dirichlet walls this is the
   lid-driven cavity of test/lid on a refined tree. */
int  gfship_tree_set_bc_u (gfship_tree * tree, int c, int d, int kind);
int  gfship_tree_set_viscosity (gfship_tree * tree, int c, double nu);
/* GfsSource {} U|V|W g with a constant intensity on a tree (as gfship_sim_set_source: MAC source of the face
   values, centred source of the update, acceleration scale of the CFL condition) */
int  gfship_tree_set_source (gfship_tree * tree, int c, double g);
gfship_multilevel_params * gfship_tree_diffusion_params (gfship_tree * tree, int c);
int  gfship_tree_start (gfship_tree * tree);
int  gfship_tree_step (gfship_tree * tree);
/* the derived variable `Divergence' (gfs_divergence, src/fluid.c:2357-2376, with
   gfs_face_interpolated_value_generic :2200-2221) of the leaves into GFSHIP_TREE_DIV */
int  gfship_tree_divergence (gfship_tree * tree);
/* host-side self-check of the plans of a tree, needs no device (CPU tests): for every level the relax
   loop of nrelax sweeps is run on the host (a) as the reference's program -- ghost copies, the cells in
   tree order through the stencil code that walks the tree --, (b) through the compiled stencils by
   dependency level, each level backwards, (c) through the plan of the whole loop, each level
   backwards; stats[0] = cell updates, [1] = dependency levels sweep after sweep, [2] = levels of the
   loop plans, [3] = number of values that differ between (a), (b), (c): 0 for valid plans */
int  gfship_tree_host_check (int dim, gfship_refine_fn refine, void * ctx, const int * side,
			     unsigned nrelax, long long stats[4]);
/* diagnostics: the cells of the sweep of gfs_relax on `level' (the cells of the level and the
   coarser leaves) and the number of dependency levels its tree order leaves on the device */
int  gfship_tree_sweep_levels (const gfship_tree * tree, int level, int * ncells, int * nlevels);

/* ---- instrumentation -------------------------------------------------------------------- */

/* time (HIP events on the domain's stream) of `reps` back-to-back sweeps of gfship_relax on
   `level`, in milliseconds per sweep; used by bench.py for the roofline entry */
int  gfship_time_relax (gfship_domain * dom, unsigned d, int level, gfship_field u,
			gfship_field rhs, gfship_field dia, int reps, double * ms_per_sweep);

/* the same for a whole relax loop of `nrelax` sweeps with the homogeneous BC between them
   (src/poisson.c:1070-1089): milliseconds per loop of the sweep kernel(s) alone (pack / unpack of
   the level and the re-arming memsets are outside the timed region); *fused is set to 1 when the
   sweeps ran pipelined in one launch */
int  gfship_time_relax_loop (gfship_domain * dom, int level, gfship_field u, gfship_field rhs,
			     gfship_field dia, unsigned nrelax, int reps, double * ms_per_loop,
			     int * fused);

/* the same, and in *ms_inclusive the time of everything such a loop costs inside a V-cycle: BC
   kernel, copy into the skewed layout, arming of the hand-off granules, the sweeps, the ghost planes
   of the last BC application, copy back (HIP events on the domain's stream around all of it) */
int  gfship_time_relax_loop_inclusive (gfship_domain * dom, int level, gfship_field u, gfship_field rhs,
				       gfship_field dia, unsigned nrelax, int reps,
				       double * ms_per_loop, int * fused, double * ms_inclusive);

#ifdef __cplusplus
}
#endif
#endif /* GFSHIP_H */
