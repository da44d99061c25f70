/* gfs_oracle.h -- CPU restatement ("oracle") of the Gerris projection / advection /
 * particle hot path on uniform single-box grids.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py may load it.  The product (libgfship) never links, loads or calls it.
 *
 * It restates, statement by statement and in the reference's traversal order, the algorithm
 * of the reference (VatsalSy/gerris-fft-particles, Gerris 1.3.2) for uniform grids:
 *   src/ftt.c, src/ftt_internal.c   traversal order, child coordinates, face traversal
 *   src/poisson.c                   coefficients, relax, residual, V-cycle, solve, diffusion
 *   src/fluid.c                     face gradients, centred gradients, divergence, norms,
 *                                   corner interpolation
 *   src/advection.c                 Godunov face values, upwinding, fluxes
 *   src/timestep.c                  projections, predictor, advection-diffusion drivers
 *   src/boundary.c, src/domain.c    BCs (periodic, symmetry, Dirichlet, Neumann), norms, CFL,
 *                                   locate, RK2 point advection
 *   src/simulation.c                simulation_run loop order, time-step selection
 *   src/particle.c, modules/particulatecommon.c   tracer step, periodic wrap
 * Each function cites the file:line it follows.  The reference itself cannot be built here
 * (needs GTS >= 0.7.4) so parity is pinned on the reference's own golden files
 * (test/poisson/res-7.ref, error.ref; test/reynolds/div*.ref; test/periodic/r0.ref; ...).
 *
 * Storage: every level l of the (implicit) quad/octree is a dense array of (n+2)^dim doubles,
 * n = 2^l, x fastest, one ghost layer per side (the reference's boundary "ghost trees").
 * Index of cell (i,j,k), 1 <= i,j,k <= n:  i + (n+2)*(j + (n+2)*k)   (k = 0 and no z in 2-D).
 * j grows with y and k with z; the reference's child ordering (x:+ for bit0, y:- for bit1,
 * z:- for bit2, ftt.c:301-316) is reproduced by the traversal lists, not by the layout.
 *
 * Compile with -ffp-contract=off: the reference is plain C compiled for x86-64 without FMA.
 */
#ifndef GFS_ORACLE_H
#define GFS_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO_MAXLEVEL 12

/* what sits behind each side of the box (reference: box->neighbor[d], boundary.h:319-327) */
enum { GO_SIDE_PERIODIC = 0,   /* GfsBoundaryPeriodic on a self edge (boundary.c:1704-1760)      */
       GO_SIDE_BOUNDARY = 1,   /* GfsBoundary with per-variable GfsBc (default: symmetry)        */
       GO_SIDE_EXTERNAL = 2 }; /* GfsBoundaryMpi: ghosts filled by the caller's exchange hook    */

/* per-variable boundary condition on a GO_SIDE_BOUNDARY side (boundary.c:45-74,253-360) */
enum { GO_BC_SYMMETRY = 0, GO_BC_DIRICHLET = 1, GO_BC_NEUMANN = 2 };

/* directions, ftt.h:78-89 */
enum { GO_RIGHT = 0, GO_LEFT, GO_TOP, GO_BOTTOM, GO_FRONT, GO_BACK };

typedef struct {           /* GfsNorm, fluid.h / fluid.c:2107-2171 */
  double bias, first, second, infty, w;
} GoNorm;

typedef struct {           /* GfsMultilevelParams, poisson.h:39-52 (same fields, same order)   */
  double tolerance;
  unsigned nrelax, erelax;
  unsigned minlevel;
  unsigned nitermax, nitermin;
  unsigned dimension;
  unsigned niter;
  unsigned depth;
  int weighted, function;
  double beta, omega;
  GoNorm residual_before, residual;
} GoMultilevelParams;

typedef struct {           /* the subset of GfsAdvectionParams used here, advection.h:50-69 */
  double cfl, dt;
  int gradient;            /* 0: gfs_center_gradient, 1: gfs_center_van_leer_gradient */
  int gc;                  /* 1: separate centred-gradient variables g[] (default for sim) */
} GoAdvectionParams;

typedef struct GoDomain GoDomain;

typedef struct {           /* one variable: all levels, ghosts included */
  GoDomain * dom;
  double * lev[GO_MAXLEVEL + 1];
  int bc[6];               /* GO_BC_* per direction, used on GO_SIDE_BOUNDARY sides */
  double * bcval[6];       /* leaf-level face-centre values (n^(dim-1)) or NULL (= 0.) */
  int component;           /* vector component 0..2, or -1 for scalars (symmetry sign) */
  int depth;               /* copy of dom->depth (lets a field outlive its domain safely) */
} GoField;

/* exchange hook for GO_SIDE_EXTERNAL sides: must fill the ghost layer of `a` (level `level`)
   on every external side from the neighbouring boxes' interior cells */
typedef void (* GoExchangeFunc) (void * ctx, double * a, int level, int kind);
/* MPI_Allreduce over the boxes: op 0 sum, 1 max, 2 min (domain.c:2135-2166, utils.h:36-42) */
typedef void (* GoReduceFunc) (void * ctx, double * vals, int n, int op);

struct GoDomain {
  int dim, depth;
  int side[6];
  int n[GO_MAXLEVEL + 1];
  size_t size[GO_MAXLEVEL + 1];     /* (n+2)^dim */
  ptrdiff_t off[GO_MAXLEVEL + 1][6];/* index offset of the neighbour in direction d */
  int * order[GO_MAXLEVEL + 1];     /* cells of a level in the reference's traversal order */
  int * border[GO_MAXLEVEL + 1][6]; /* cells adjacent to side d, in traversal order */
  int nborder[GO_MAXLEVEL + 1];
  double * w[6][GO_MAXLEVEL + 1];   /* GFS_STATE(cell)->f[d].v used as Poisson weights */
  GoExchangeFunc exchange;
  void * exchange_ctx;
  GoReduceFunc reduce;
  void * reduce_ctx;
  int mpi_order;                    /* 1: boundary-cells-first sweep order (domain.c:1093-1125) */
};

/* ---- core (go_core.c) ---- */
GoDomain * go_domain_new (int dim, int depth, const int side[6]);
void       go_domain_destroy (GoDomain * dom);
size_t     go_level_size (const GoDomain * dom, int level);
size_t     go_index (const GoDomain * dom, int level, int i, int j, int k);
const int *go_order (const GoDomain * dom, int level);
GoField *  go_field_new (GoDomain * dom, int component);
void       go_field_destroy (GoField * f);
double *   go_field_level (GoField * f, int level);
void       go_field_set_bc (GoField * f, int d, int type, const double * val);
void       go_bc (GoField * v, GoField * v1, int level);              /* gfs_domain_copy_bc */
void       go_homogeneous_bc (GoField * ov, GoField * v, int level);  /* gfs_domain_homogeneous_bc */
void       go_domain_set_hooks (GoDomain * dom, GoExchangeFunc ex, void * ex_ctx,
				GoReduceFunc red, void * red_ctx);
void       go_domain_set_overlap (GoDomain * dom, int overlap);
double *   go_domain_weight (GoDomain * dom, int d, int level);
size_t     go_snapshot_tree_bytes (const GoDomain * dom, int nvars);
size_t     go_snapshot_tree_write (const GoDomain * dom, int nvars, GoField ** f, unsigned char * buf);
void       go_cell_pos (const GoDomain * dom, int level, int i, int j, int k, double pos[3]);

/* ---- Poisson (go_poisson.c) ---- */
void   go_multilevel_params_init (GoMultilevelParams * par, int dim);
void   go_poisson_coefficients (GoDomain * dom);                      /* alpha = NULL (unity) */
void   go_poisson_coefficients_alpha (GoDomain * dom, double * const alpha[3]);
void   go_relax (GoDomain * dom, unsigned d, int level, double omega,
		 GoField * u, GoField * rhs, GoField * dia);
void   go_residual (GoDomain * dom, unsigned d, int level,
		    GoField * u, GoField * rhs, GoField * dia, GoField * res);
GoNorm go_norm_residual (GoDomain * dom, double dt, GoField * res);
void   go_poisson_cycle (GoDomain * dom, GoMultilevelParams * p,
			 GoField * u, GoField * rhs, GoField * dia, GoField * res);
void   go_poisson_solve (GoDomain * dom, GoMultilevelParams * par,
			 GoField * lhs, GoField * rhs, GoField * res, GoField * dia, double dt);
GoNorm go_norm_variable (GoDomain * dom, GoField * v);
void   go_relax_lexicographic (GoDomain * dom, unsigned d, int level, double omega,
			       GoField * u, GoField * rhs, GoField * dia);

#ifdef __cplusplus
}
#endif
#endif /* GFS_ORACLE_H */
