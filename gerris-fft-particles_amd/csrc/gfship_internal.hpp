// gfship_internal.hpp -- shared declarations of libgfship (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <vector>
#include <deque>
#include <string>
#include "gfship.h"

namespace gfship {

// Per-level dense layout (replaces FttOct pointers, src/ftt.h:134-159): (n+2)^(dim-1) rows of
// px doubles.  Interior cell i = 1 sits at x index xo + 1 = 2 so that rows of interior cells
// start 16-byte aligned (px is even); ghosts at xo + 0 and xo + n + 1.
struct Layout {
  int n = 0;        // cells per side
  int px = 0;       // x pitch in doubles
  int xo = 1;       // x offset of i = 0
  int rows = 0;     // n + 2
  int dim = 3;
  long sy = 0;      // stride of j  (= px)
  long sz = 0;      // stride of k  (= px*rows, 0 in 2-D)
  size_t total = 0; // doubles per field-level
  __host__ __device__ inline long idx (int i, int j, int k) const {
    return xo + i + sy*j + sz*k;
  }
};

struct Field {
  bool used = false;
  int component = -1;
  double * lev[GFSHIP_MAXLEVEL + 1] = {};
  int bc[6] = {0, 0, 0, 0, 0, 0};
  double * bcval[6] = {};      // device, leaf-level face values or nullptr
  // level is known to hold only zeros (set by alloc / fill (0.), cleared by every other writer
  // of a *named* field: uploads and kernels that take the Field as output)
  bool zero[GFSHIP_MAXLEVEL + 1] = {};
  // set by a caller that wants the non-leaf levels recomputed (gfs_cell_coarse_init,
  // src/adaptive.c:43-58) the next time one of them is read or written (coarse_flush); the
  // simulation loop does not use it: its coarse values are a snapshot taken in mid-step
  bool coarse_stale = false;
  // the non-leaf levels have been given values (gfs_cell_coarse_init, an upload, a snapshot)
  bool coarse_valid = false;
};

// scratch of the skewed exact-order sweep of one level (relax_skew.hip)
struct SkewPlan {
  int ntj = 0, RT = 0;
  size_t hb_words = 0;
  double * us = nullptr, * rs = nullptr, * ds = nullptr;
  double * ws[6] = {};            // skewed face weights of the level (weighted relax), packed when ws_stamp is stale
  unsigned long ws_stamp = 0;
  void * hb = nullptr;            // hand-off granules (J side then K side)
  void * hbf = nullptr;           // granules of the fused relax loop (relax_skew_loop.hip): two sets
  // the sets are used in turn: a loop kernel of relax_patch_loop.hip arms the OTHER set for the next
  // loop of the level while its tiles wait for their first hand-off (armed = granules armed from
  // the start of the set, 0 = to be armed by a fill before the launch)
  bool rs_ready = false;          // the skewed rhs already holds the residual of this cycle (launch_restrict_pack)
  int cur_set = 0;
  size_t armed[2] = { 0, 0 };
  unsigned * arm_cum = nullptr;   // share of each tile in that arming (cumulative weights)
  // the set the next loop of the level will use is armed ahead of time on the domain's side stream,
  // beside the coarser levels of the cycle (skew_arm_ahead): the loop then waits for this event
  hipEvent_t arm_ev = nullptr;
  bool arm_wait = false;
  bool loop_checked = false;      // the trial run of the fused loop has been made on this level
  void * stats_loop = nullptr;    // optional per-tile, per-sweep timing of the fused loop (debug)
  void * ctl = nullptr;           // { ticket, err }
  unsigned long long * stats = nullptr; // optional per-tile timing (debug)
  unsigned short * order = nullptr;
  unsigned short * xorder = nullptr;   // tiles by XCD block (relax_skew_loop.hip)
};

// cell update of the exact-order sweeps: kind 0 = Poisson relax (unit weights), kind 1 =
// diffusion_relax with the level's uniform face weight w and h2 = h*h
struct RelaxOp {
  int kind = 0;                  // 2: Poisson relax with the face weights f[d].v of the level (wf)
  double w = 1., h2 = 1.;
  const double * wf[6] = {};
};

// device-side description of the six sides for the BC kernel
struct BcDesc {
  int side[6];
  int type[6];
  const double * val[6];
  int component;
  int homogeneous;
};

} // namespace gfship

struct gfship_domain {
  int dim = 3, depth = 0, device = 0;
  int side[6] = {1, 1, 1, 1, 1, 1};
  int relax_mode = GFSHIP_RELAX_EXACT;
  gfship::Layout lay[GFSHIP_MAXLEVEL + 1];
  std::deque<gfship::Field> fields;   // deque: handles stay valid while fields are added
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // side stream of the V-cycle: byte-moving work that nothing on the main stream waits for soon (the
  // arming of the hand-off granules of the next relax loop of a fine level) runs here, beside the
  // latency-bound coarse levels, forked and joined with events
  hipStream_t side_stream = nullptr;
  hipEvent_t side_fork = nullptr;
  bool no_arm_ahead = false;       // GFSHIP_NO_ARM_AHEAD=1: the granules are armed in line, before the loop
  // called before a caller overwrites a field of the domain (upload, snapshot read): a simulation that
  // keeps derived state unstored (its MAC velocities, simulation.hip: materialize_un) stores it first
  int (* before_write) (void *) = nullptr;
  void * before_write_ctx = nullptr;
  double * d_scratch = nullptr;   // reduction scratch
  size_t scratch_doubles = 0;
  double * h_pinned = nullptr;    // pinned host buffer for small read-backs
  gfship_exchange_fn exchange = nullptr; void * exchange_ctx = nullptr;  // GfsBoundaryMpi hooks
  gfship_reduce_fn reduce = nullptr;     void * reduce_ctx = nullptr;
  bool has_external = false;
  void * comm = nullptr;          // in-library RCCL transport (transport.hip), replaces the hooks
  int overlap = 0;                // the domain parameter `overlap' of a parallel run (src/domain.c:225,682)
  // the lattice of boxes this box belongs to (gfship_domain_comm_init or gfship_domain_set_gather)
  int lat_rank = -1, lat_n = 0, lat_b[3] = { 1, 1, 1 };
  gfship_gather_fn gather = nullptr; void * gather_ctx = nullptr;
  // the coarse end of a V-cycle computed for every box of the lattice on every rank
  // (lattice_cycle_kernel, poisson_kernels.hip): gathered residuals, layers between the boxes, barrier
  double * lat_res = nullptr, * lat_xch = nullptr;
  unsigned * lat_bar = nullptr;
  size_t lat_res_doubles = 0, lat_xch_doubles = 0;
  double * gfv_send[6] = {}, * gfv_recv[6] = {};   // states beyond the MPI sides of the tiled Godunov kernels
  double src[3] = { 0., 0., 0. };       // GfsSource {} U/V/W: constant intensities (gfship_sim_set_source)
  bool no_xcd_scope = true;             // GFSHIP_XCD_SCOPE=1: XCD blocks of tiles + narrower-scope stores towards same-XCD consumers (measured: no gain)
  bool no_fused_restriction = false;    // GFSHIP_NO_FUSED_RESTRICTION=1: restrict_kernel, then the copy of the rhs
  bool no_kernel_arming = true;         // GFSHIP_KERNEL_ARMING=1: the loop kernels arm the other granule set (measured: no gain)
  bool no_fused_prolongation = false;   // GFSHIP_NO_FUSED_PROLONGATION=1: prolongate_kernel, then the copy
  bool no_fused_mpi = false;      // GFSHIP_NO_FUSED_MPI=1: face-value arrays on boxes with MPI sides
  unsigned long long n_lattice_cycles = 0, n_fused_mpi = 0;   // gfship_domain_path_counts
  bool lattice_attr_set = false;  // dynamic-LDS limit of lattice_cycle_kernel raised
  bool no_lattice_cycle = false;  // GFSHIP_NO_LATTICE_CYCLE=1: one exchange per sweep on every level
  void * mpi_plan[GFSHIP_MAXLEVEL + 1] = {};  // MPI-sides-first sweep order of each level (poisson_kernels.hip)
  gfship_field dp_cache = -1;     // the `dp` temporary of gfs_poisson_cycle, kept between cycles
  int skew_resident = -1;         // workgroups of the fused relax loop that fit on the device
  int patch_resident = 0;         // the same for the kernels of relax_patch_loop.hip
  bool wave_loop = false;         // fused relax loops by the experimental one-wave-per-tile kernel (GFSHIP_WAVE_LOOP=1)
  bool xcd_place = false;         // XCD-aware tile placement in the loop kernel (experiment, GFSHIP_XCD_PLACE=1)
  bool skew_old = false;          // single sweeps by the older four-wave kernel (GFSHIP_SKEW_OLD)
  bool patch = true;              // 2 x 2 lines per lane (relax_patch_loop.hip) on the levels where it wins; GFSHIP_SKEW_LINES=1: one line per thread everywhere
  int patch_min_n = 128;          // ... i.e. n >= 128 (GFSHIP_PATCH_MIN_N): on 64^3 and 32^3 (16 and 4 tiles) one line per thread is a few us faster
  bool no_fused_godunov = false;  // face-value arrays + separate kernels even on periodic boxes
  bool no_fused_godunov3 = false; // one launch per velocity component instead of the three at once (GFSHIP_NO_ADVECT3)
  bool no_fused_loop = false;     // one launch per sweep even where the fused loop applies
  bool force_hyperplane = false;  // debug/bench: per-hyperplane launches instead of relax_skew
  bool unit_weights = false;      // gfship_poisson_coefficients called with alpha = NULL
  bool weighted = false;          // gfship_poisson_coefficients_alpha: the face weights live in wf[]
  unsigned long weights_stamp = 0; // bumped whenever wf[] is recomputed (skewed copies are then stale)
  gfship_field wf[6] = {-1, -1, -1, -1, -1, -1};   // GFS_STATE (cell)->f[d].v as Poisson weights, all levels
  gfship::SkewPlan skew[GFSHIP_MAXLEVEL + 1];
  double diff_w[GFSHIP_MAXLEVEL + 1] = {};  // diffusion face weight of each level
  double * cfl_partial = nullptr;   // per-block max |un|, |u| of the fused projection update
  size_t cfl_nblocks = 0, cfl_used = 0;
  bool cfl_dirty = false;
  bool coarse_attr_set = false;   // dynamic-LDS limit of coarse_cycle_kernel raised
  gfship_field res_cache = -1;    // the `res` temporary of gfs_diffusion
  bool diff_ready = false;        // gfship_diffusion_coefficients called
};

namespace gfship {

void set_error (const char * fmt, ...);
int  hip_fail (hipError_t e, const char * what, const char * file, int line);

#define GFSHIP_HIP(call) do { hipError_t e_ = (call); \
    if (e_ != hipSuccess) return gfship::hip_fail (e_, #call, __FILE__, __LINE__); } while (0)

// wait for the stream by polling it: the small results the host needs in the middle of a step
// (norms, CFL) come back a few microseconds sooner than through a blocking synchronisation
inline hipError_t stream_wait_spin (hipStream_t st)
{
  hipError_t e;
  while ((e = hipStreamQuery (st)) == hipErrorNotReady)
    ;
  return e;
}

#define GFSHIP_CHECK(cond, code, ...) do { if (!(cond)) { \
    gfship::set_error (__VA_ARGS__); return (code); } } while (0)

Field * get_field (gfship_domain * dom, gfship_field f);
int coarse_flush (gfship_domain * dom, Field * F, int level);
inline long ncells (const Layout & L) {
  return L.dim == 3 ? (long) L.n*L.n*L.n : (long) L.n*L.n;
}

// kernels launchers (poisson_kernels.hip, bc.hip)
int launch_bc (gfship_domain * dom, Field * v, Field * v1, int level, int homogeneous);
int call_exchange (gfship_domain * dom, double * ptr, int level, int kind);
int bc_mpi_begin (gfship_domain * dom, Field * v1, int level);
int bc_mpi_end (gfship_domain * dom, Field * v, Field * v1, int level, int homogeneous);
int call_reduce (gfship_domain * dom, double * vals, int n, int op);
int call_reduce_norm (gfship_domain * dom, double * sums, int nsum, double * mx);
// transport.hip
int comm_exchange (gfship_domain * dom, double * a, int level, int kind);
int comm_exchange_raw (gfship_domain * dom, double * const send[6], double * const recv[6], size_t count);
int comm_migrate (gfship_domain * dom, int rs, const int nsend[6], const double * const send[6],
		  int nrecv[6], std::vector<double> recv[6]);
int comm_allgather (gfship_domain * dom, const double * send, double * recv, size_t count);
int multi_buffers (gfship_domain * dom);
int launch_bc_multi (gfship_domain * dom, Field * const * v, int nf, int level, int homogeneous);
int call_gather (gfship_domain * dom, const double * send, double * recv, size_t count);
int comm_exchange_begin (gfship_domain * dom, double * a, int level);
int comm_exchange_end (gfship_domain * dom, double * a, int level);
int comm_reduce (gfship_domain * dom, double * sums, int nsum, double * maxs, int nmax,
		 double * mins, int nmin);
void comm_free (gfship_domain * dom);
int launch_relax_exact (gfship_domain * dom, unsigned dimension, int level, double omega,
			double * u, const double * rhs, const double * dia,
			const RelaxOp * op = nullptr);
int launch_relax_mpi_first (gfship_domain * dom, unsigned dimension, int level, double omega,
			    double * u, const double * rhs, const double * dia, const RelaxOp * op,
			    int (* after_shell) (void *), void * ctx);
void mpi_order_free (gfship_domain * dom);
int launch_relax_loop_small (gfship_domain * dom, unsigned dimension, int level, double omega,
			     Field * dp, Field * ubc, const double * rhs, const double * dia,
			     unsigned nrelax, bool * done, const RelaxOp * op = nullptr);
int coarse_cycle_top (gfship_domain * dom, int minlevel);
int lattice_cycle_top (gfship_domain * dom, int minlevel, Field * dia);
int lattice_check_error (gfship_domain * dom);
int launch_lattice_cycle (gfship_domain * dom, unsigned dimension, double omega, int lmin, int ltop,
			  const unsigned * nrelax, Field * dp, Field * ubc, Field * res);
int launch_coarse_cycle (gfship_domain * dom, unsigned dimension, double omega, int lmin, int ltop,
			 const unsigned * nrelax, Field * dp, Field * ubc, Field * res, Field * dia);
int launch_relax_redblack (gfship_domain * dom, unsigned dimension, int level, double omega,
			   double * u, const double * rhs, const double * dia);
int launch_residual (gfship_domain * dom, int level, const double * u, const double * rhs,
		     const double * dia, double * res);
int launch_residual_weighted (gfship_domain * dom, int level, const double * u, const double * rhs,
			      const double * dia, double * res);
int launch_poisson_weights (gfship_domain * dom, double * const alpha[3]);
RelaxOp weighted_op (gfship_domain * dom, int level);
int launch_restrict (gfship_domain * dom, unsigned dimension, int level_coarse, double * v_coarse,
		     const double * v_fine);
int launch_prolongate (gfship_domain * dom, int level_coarse, const double * v_coarse,
		       double * v_fine);
int launch_correct (gfship_domain * dom, int level, double * u, const double * dp);
int launch_fill (gfship_domain * dom, int level, double * a, double value);
int launch_residual_norm (gfship_domain * dom, int level, const double * u, const double * rhs,
			  const double * dia, double * res, double scale, double weight,
			  double * out /* 5 values, or nullptr: left in h_pinned[8..12], not waited for */, bool dia_zero = false);
int launch_norm_async (gfship_domain * dom, int level, const double * a, double scale, double weight);
int launch_norm (gfship_domain * dom, int level, const double * a, double scale, double weight,
		 double out[5] /* bias(sum of scaled), first, second, infty, raw sum */);

// timestep_kernels.hip
int launch_face_interp_un (gfship_domain * dom, double * const u[3], double * const un[3]);
int launch_divergence (gfship_domain * dom, double * const un[3], double * div, double dt);
int launch_correct_un (gfship_domain * dom, const double * p, double * const un[3], double dt);
int launch_centered_gradient (gfship_domain * dom, const double * p, double * const g[3]);
int launch_correct_centered (gfship_domain * dom, double * const u[3], double * const g[3], double dt);
int launch_correct_centered_coarse (gfship_domain * dom, double * const u[3], double * const g[3],
				    double dt, double * const uc[3]);
int launch_coarse_init_from (gfship_domain * dom, Field * const * v, int nf, int top);
int launch_coarse_init_levels (gfship_domain * dom, Field * const * v, int nf, int top, int bottom);
int launch_advected_face_values (gfship_domain * dom, const double * v, double * const u[3],
				 double * const un[3], double dt, int use_centered, int gradient,
				 double * const fv[6], int cmask, double visc = 0., double gsrc = 0.);
int launch_face_bc (gfship_domain * dom, Field * v, double * const fv[6], int cmask);
int launch_predict_un (gfship_domain * dom, int cc, const double * uc, double * const fv[6],
		       double * unc);
int launch_flux_update (gfship_domain * dom, bool velocity, double * v, double * const un[3],
			double * const fv[6], const double * gm, const double * gc, double dt,
			double gsrc = 0.);
int launch_cfl (gfship_domain * dom, double * const u[3], double * const un[3],
		const double visc[3], double * cfl2);
int launch_coarse_init (gfship_domain * dom, Field * const * v, int nf);
int launch_velocity_divergence (gfship_domain * dom, double * const u[3], double * out);
bool godunov_fused_supported (const gfship_domain * dom);
bool godunov_fused_mpi_supported (const gfship_domain * dom);
int launch_project_correct (gfship_domain * dom, const double * p, double * const un[3],
			    double * const g[3], double * const u[3], double dt, bool want_max);
int launch_cfl_from_max (gfship_domain * dom, double * cfl2);
int launch_face_interp_div (gfship_domain * dom, double * const u[3], double * const un[3] /* or nullptr */,
			    double * div, double dt);
int launch_project_correct_weighted (gfship_domain * dom, const double * p, double * const un[3],
				     double * const g[3], double * const u[3], double dt);
int launch_project_correct_lazy (gfship_domain * dom, const double * p, double * const u[3],
				 double * const g[3], double * const uo[3], double dt);
// div != nullptr: the caller's next operation is the MAC projection with time step div_dt; where the
// predictor runs as the sweep along z it also leaves the scaled divergence of the new face velocities
// in div (*div_done set): mac_projection then skips its divergence pass
int launch_predict_un_fused (gfship_domain * dom, double * const u[3], double dt, int gradient,
			     const double visc[3], double * const un[3], double * div = nullptr,
			     double div_dt = 0., bool * div_done = nullptr);
// corr_dt != 0: gfs_correct_centered_velocities with gc and corr_dt applied in the same pass, and with uc
// the level below the leaves of the corrected velocities (gfs_cell_coarse_init's first level)
int launch_advect3_fused (gfship_domain * dom, double * const v[3], double * const out[3],
			  double * const un[3], double * const gm[3], double * const gc[3],
			  double dt, int gradient, double corr_dt = 0., double * const uc[3] = nullptr);
int launch_advect_fused (gfship_domain * dom, bool velocity, const double * v, double * out,
			 double * const un[3], const double * gm, const double * gc, double dt,
			 int gradient, double visc, double gsrc = 0.);


// relax_skew.hip
bool skew_supported (const gfship_domain * dom, int level);
int  launch_relax_loop_skew (gfship_domain * dom, int level, Field * dp, Field * ubc,
			     const double * rhs, const double * dia, bool dia_zero,
			     unsigned nrelax, bool bc, double * correct_into = nullptr,
			     const double * prolong_from = nullptr, const RelaxOp * op = nullptr);
void skew_free (gfship_domain * dom);
bool skew_loop_supported (gfship_domain * dom, int level, unsigned nrelax, bool bc);
int  skew_loop_run (gfship_domain * dom, int level, SkewPlan * S, double * u_nat, bool has_dia,
		    unsigned nrelax, float * ms = nullptr, const Field * ubc = nullptr,
		    const RelaxOp * op = nullptr);
int  skew_time_loop (gfship_domain * dom, int level, Field * u, const double * rhs,
		     const double * dia, bool dia_zero, unsigned nrelax, int reps,
		     double * ms_per_loop, int * fused, double * ms_inclusive = nullptr);
int  skew_check_error (gfship_domain * dom);
// arm the granules of the next relax loop of `level' on the side stream, from this point of the main stream
int  skew_arm_ahead (gfship_domain * dom, int level, unsigned nrelax);
// relax_patch_loop.hip
int  patch_resident_per_cu ();
// the prolongation onto `level' can be done by the copy into the skewed layout of its relax loop
bool prolongation_fused (gfship_domain * dom, unsigned dimension, int level, unsigned nrelax);
/* weighted sweeps (face weights from alpha) run on the six-wave kernel with its tile-skewed layout */
inline bool patch_level (const gfship_domain * dom, int level) { return dom->patch && !dom->weighted && dom->lay[level].n >= dom->patch_min_n; }
int  patch_pack (gfship_domain * dom, int level, SkewPlan * S, const double * u, const double * rhs,
		 const double * dia, const double * coarse = nullptr);
int  patch_unpack (gfship_domain * dom, int level, SkewPlan * S, double * u, double * add_into);
int  patch_restrict_pack (gfship_domain * dom, int level, SkewPlan * S, const double * res,
			  double * res_coarse, SkewPlan * Sc, unsigned dimension);
// restriction of the residual of `level' onto level - 1 together with the copy of the residual into
// the skewed layout of the relax loop of `level' (and of level - 1 where that is a 2 x 2 level too)
int  launch_restrict_pack (gfship_domain * dom, unsigned dimension, int level, Field * res,
			   bool coarse_fused);
void skew_dump_stats (gfship_domain * dom, int level);
int  skew_time_sweeps (gfship_domain * dom, int level, Field * u, const double * rhs,
		       const double * dia, bool dia_zero, int reps, double * ms_per_sweep);

} // namespace gfship
