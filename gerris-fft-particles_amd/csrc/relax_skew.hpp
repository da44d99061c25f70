// relax_skew.hpp -- definitions shared by the pipelined exact-order sweep kernels
#pragma once
#include "gfship_internal.hpp"

namespace gfship {

#define SK_T   16            /* tile edge (lines) */
#define SK_NL  (SK_T*SK_T)   /* lines = threads per tile */
#define SK_PAD (2*SK_T - 2)  /* extra rows of a tile: max skew */
#ifndef SK_D
#define SK_D   6             /* prefetch distance (steps) of the own streams: 16 was 12 % slower at */
                             /* 256^3 (round 2, tools/lab/knobs.sh: 4, 6, 8, 12, 16 tried): what is */
                             /* in flight in a CU is what a hand-off poll of that CU queues behind  */
#endif
#ifndef SK_DH
#define SK_DH  3             /* prefetch distance of the halo streams (divides SK_D): the lag */
                             /* between neighbouring tiles grows with it                     */
#endif
/* rows of padding in front of and behind every tile, so that prefetch addresses never need
   clamping: 15 rows in front (first lines of the next tile are read at row t - 15), 2*SK_D behind
   (rows up to T + SK_D, T rounded up to SK_D) */
#define SK_FP  48
/* rows of a tile's hand-off / snapshot granule array: n + 30 used, the streams read ahead by up
   to SK_D (rounding of T) + SK_DH + 1 rows */
#define SK_HROWS(n_) ((n_) + 2*SK_T + 36)   /* 36 = 16 + 4 + 16: the largest read-ahead of any of the kernels */

typedef unsigned long long u64;
#define SK_SENTINEL 0xFFFFFFFFFFFFFFFFull

struct SkewArgs {
  Layout L;
  int ntj;                 // tiles per side
  int RT;                  // rows per tile
  double * us;             // skewed u
  const double * rs;       // skewed rhs
  const double * ds;       // skewed dia (or nullptr)
  double * un;             // natural u (ghost layer + mirrored side cells)
  u64 * hbJ;               // [tile][n + SK_T - 1][SK_T] new values of line a = 15
  u64 * hbK;               // [tile][n + SK_T - 1][SK_T] new values of line b = 15
  const unsigned short * order; // ticket -> tile (anti-diagonal major)
  unsigned * ticket;       // ticket counter (zeroed before the launch)
  unsigned * err;          // set to 1 when a bounded spin gives up
  const u64 * dummy;       // 8 readable bytes for the streams a lane does not need
  u64 * stats;             // optional per-tile { start, end, spins, slow entries } (debug)
};

#define SK_MAXF 8    /* sweeps per launch */
#ifndef SK_POLL_SLEEP
#define SK_POLL_SLEEP 0   /* s_sleep argument between two polls of a hand-off granule */
#endif
#ifndef SK_EXP
#define SK_EXP 0    /* timing experiments: 1 plain halo prefetch loads, 2 plain granule stores */
#endif
#define SK_HLOAD(p_) ((SK_EXP & 1) ? *(p_) : load_sc1 (p_))
#ifndef SK_KO
#define SK_KO 0      /* timing experiments only: knock out parts of the step (wrong results) */
#endif


struct SkewLoopArgs {
  Layout L;
  int ntj, RT, nsweeps;
  // homogeneous BC of the sides d = 0..5 (right, left, top, bottom, front, back) between the sweeps
  // of a fused loop: sgn[d] = 0 periodic; otherwise ghost = sgn[d] * adjacent interior value (-1
  // Dirichlet and the normal component at a symmetry side, +1 Neumann and symmetry otherwise)
  double sgn[6];
  int mirror;              // single sweep with the BC kernel around it: cells next to the box sides
                           // are also written to the natural array (any kind of side)
  double * us;             // skewed u (in place)
  const double * rs;       // skewed rhs
  const double * ds;       // skewed dia (or nullptr)
  double * un;             // natural u: ghosts of sweep 0 are read, ghosts of the last BC written
  u64 * hb;                // per sweep: [J hand-off | K hand-off | J snapshot | K snapshot]
  long hb_sweep;           // granules per sweep
  long hb_words;           // granules of one hand-off array (ntiles*hstride)
  const unsigned short * order;
  unsigned * ticket, * err;
  const u64 * dummy;
  u64 * stats;             // optional [tile][sweep]{start, end} (debug, GFSHIP_SKEW_STATS)
  // XCD-aware placement (all tiles resident): the tiles are split into 8 blocks, one per XCD, and
  // a workgroup claims a tile of the block of the XCD it runs on (any other block once its own is
  // exhausted): most hand-offs then stay inside one L2
  const unsigned short * xorder;   // [8][per_xcd] tiles of each block, anti-diagonal order
  unsigned * xticket;              // [8] ticket counters (zeroed before the launch)
  int per_xcd;                     // 0: placement by the single ticket counter
  // arming of the other granule set (relax_patch_loop.hip): arm_pairs 16-byte pairs from `arm',
  // shared between the tiles by arm_cum[tile] .. arm_cum[tile + 1] of arm_cum[ntiles]
  u64 * arm;
  unsigned long long arm_pairs;
  const unsigned * arm_cum;
  // XCD of the workgroup that claimed each tile (armed all ones with the granules; relax_patch_loop.hip)
  unsigned * tile_xcd;
  int fault_tile;          // test of the error path (GFSHIP_FAULT_DROP_HANDOFF=tile): that tile publishes nothing in sweep 0
  int near_mode;           // stores towards a consumer on the same XCD: 0 agent scope like the others, 1 plain, 2 workgroup scope
  // cell update (RelaxOp): 0 = relax (src/poisson.c:507-530, unit weights), 1 = diffusion_relax
  // (:1455-1484) with the uniform face weight w of the level and h2 = h*h; dia is then rhoc
  // 2 = relax with the six face weights f[d].v of every cell (gfs_poisson_coefficients with a
  // GfsFunction alpha): ws[d] are skewed copies of the weights, streamed beside u / rhs / dia
  int op;
  double w, h2;
  const double * ws[6];
};

typedef __attribute__((address_space(1))) u64 gu64;

__device__ __forceinline__ u64 load_sc1 (const u64 * p)
{
  return __hip_atomic_load ((gu64 *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_sc1 (u64 * p, u64 v)
{
  __hip_atomic_store ((gu64 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (bb - rhs)/aa of relax (src/poisson.c:527) when dia == 0: aa = 0. + 1. + ... + 1. = 6. exactly,
// and the correctly rounded quotient x/6 is obtained without the 14-instruction IEEE division
// sequence: q = x*r, rem = fma (-q, 6, x) (exact), q' = fma (rem, r, q) with r = RN (1/6)
// (Markstein's correction step).  x/6 = (x/2)/3 is never closer than 1/6 ulp to a rounding
// boundary while q + rem*r differs from x/6 by less than 2^-52 ulp, so q' = RN (x/6) whenever
// nothing underflows: for x = +0 the sequence gives +0 (x is never -0 here: it is a difference
// whose minuend is a sum started from +0.), large x does not overflow (q <= x/6, the product in
// the fma is exact), infinities and NaNs give NaN where the division gives inf/NaN (the solve has
// diverged either way); only for 0 < |x| < 2^-1000, where q or the remainder may be subnormal,
// the true division is used.  The guard is one exponent extraction and one integer compare, off
// the dependent chain (tools/lab/step_lab.hip: the earlier two-sided floating-point range test
// cost more than the division it replaced).  Checked against x/6. on 1.5e9 operands.
__device__ __forceinline__ double divide_by_6 (double x)
{
  const double r = 0x1.5555555555555p-3;
  const double q = x*r;
  const double rem = __builtin_fma (- q, 6., x);
  double q2 = __builtin_fma (rem, r, q);
  // frexp exponent: 0 for zeros, infinities and NaNs; below -999 only for tiny non-zero x
  const bool tiny = __builtin_amdgcn_frexp_exp (x) < -999;
  if (__builtin_expect (__builtin_amdgcn_ballot_w64 (tiny) != 0, 0))
    q2 = x/6.;
  return q2;
}

int patch_loop_launch (gfship_domain * dom, const SkewLoopArgs & A, int ntiles, bool has_dia,
		       unsigned nrelax, float * ms);

// diffusion_relax of one cell, src/poisson.c:1455-1484 with gfs_face_cm_weighted_gradient's
// same-level branch (relax_value<3, 1> of poisson_kernels.hip): d = 0..5 = right, left, top, bottom,
// front, back
__device__ __forceinline__ double diffusion_cell (double right, double left, double top, double bottom,
						  double front, double back, double rhs, double dia,
						  double w, double h2)
{
  double ga = 0., gb = 0.;
  ga += w; gb += w*right;
  ga += w; gb += w*left;
  ga += w; gb += w*top;
  ga += w; gb += w*bottom;
  ga += w; gb += w*front;
  ga += w; gb += w*back;
  const double a = dia*h2;
  ga = 1. + ga/a;
  return (gb/a + rhs)/ga;
}

} // namespace gfship
