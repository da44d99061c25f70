// relax_patch_loop.hip -- the pipelined exact-order relax loop with 2 x 2 lines per lane.
//
// Same schedule, same granule protocol and same arithmetic per cell as relax_skew_loop.hip (read
// that file first); what changes is who owns what inside a tile.  There a tile of 16 x 16 lines is
// 256 threads, one line each, and line (a, b) runs a + b steps behind line (0, 0): a tile's own
// skew is 30 steps, and every hop of the chain of tiles across the box -- which is what the loop
// waits for, tools/lab + DESIGN.md 5 -- carries 16 of them.  Here ONE wave computes the tile: lane
// (A, B), A, B = 0..7, owns the lines (2A + da, 2B + db), da, db = 0, 1, and computes their four
// cells of one I per step, in the order (0,0), (1,0), (0,1), (1,1): inside the patch the new values
// it needs are the ones it has just computed, in registers.  Lane (A, B) runs A + B steps behind
// lane (0, 0): the tile's skew is 14 steps, a hop carries 8.  A step costs about what it cost before
// (it is bounded by what one CU streams, 6 KB per step, and by the issue of one wave; the four
// SIMDs no longer meet at a barrier with a quarter of the work each), so the chain is nearly twice
// as fast.  The memory side gains too: a lane's four cells of a row are 32 contiguous bytes
// (two 16-byte accesses per array and step instead of four 8-byte ones: rows are 2 KB as before).
//
// Layout (patch-skewed): element (I, a, b) of tile T sits at
//     T*tstride + SK_FP*256 + (I + (a >> 1) + (b >> 1))*256 + 128*(b & 1) + 2*((a >> 1) + 8*(b >> 1)) + (a & 1)
// so that at step t lane (A, B) = lane l reads and writes 16 bytes at column 2 l of each half of
// row t (t + 1 for the old values one cell ahead): the lines with even b in the first KB of the row,
// those with odd b in the second, and every 16-byte access of the wave covers one contiguous KB.
//
// Workgroup = the compute wave + the halo wave + the store wave of relax_skew_loop.hip; the LDS
// exchange grids X (new values of the previous step) and Y (old values one cell ahead) are indexed
// by line as before, so the strips of the halo wave and the lines of the store wave keep their LDS
// positions; their lags in steps change with the skew:
//     line (-1, m) / (m, -1) is read by lane (0, m >> 1) / (m >> 1, 0)     at I = t - (m >> 1)
//     line (16, m) / (m, 16) is read by lane (7, m >> 1) / (m >> 1, 7)     at I = t - 7 - (m >> 1)
//     hand-off granule of cell I of line (15, m) / (m, 15): row I + (m >> 1)       (consumer's step)
//     snapshot granule of cell I of line (0, m) / (m, 0):   row I + (m >> 1) + 7
#include "relax_skew.hpp"
#include <cstdlib>
#include <vector>

#ifndef PK_D
#define PK_D 9       /* prefetch distance (steps) of the compute wave's streams (6, 8, 9, 10, 12, 15 tried) */
#endif
#ifndef PK_DH
#define PK_DH 3      /* prefetch distance of the halo strips (divides PK_D) */
#endif
#ifndef PK_KO
#define PK_KO 0      /* timing experiments only (wrong results): 1 no streaming loads, 2 no row stores, 4 no LDS publish */
#endif
#define PK_NTHREADS 192
#define PK_SKEW 14   /* largest A + B */

namespace gfship {

typedef double d2 __attribute__((ext_vector_type(2)));

// the tile's share of the arming of the other granule set (all threads of the workgroup; plain
// stores: the set is read by a later launch)
__device__ __forceinline__ void patch_arm_other_set (const SkewLoopArgs & A, int tile, int tid, int nt)
{
  if (A.arm_pairs == 0) return;
  typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
  const unsigned long long W = A.arm_cum[A.ntj*A.ntj];
  const unsigned long long lo = A.arm_pairs*A.arm_cum[tile]/W, hi = A.arm_pairs*A.arm_cum[tile + 1]/W;
  ull2 * const p = (ull2 *) A.arm;
  const ull2 ones = { ~0ull, ~0ull };
  for (unsigned long long i = lo + tid; i < hi; i += nt)
    __builtin_nontemporal_store (ones, p + i);
}

__device__ __forceinline__ unsigned patch_xcc_id ()
{
  unsigned v;
  asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s" (v));
  return v & 7;
}

// The tickets are armed with the granules: all ones, so the claims add one.  With per_xcd != 0 (all
// tiles resident) the tiles are split into 8 blocks and a workgroup claims a tile of the block of
// the XCD it runs on (any other block once its own is exhausted), and publishes the XCD it really
// runs on: the producer of a granule stream whose consumer tile runs on the same XCD writes it
// with plain stores -- they stop in the L2 the consumer's agent-scope polls hit (0.29 us on an idle
// machine, tools/lab/xcd_lab.hip) instead of going through to memory, whose latency under the load
// of 256 streaming tiles is most of what a hand-off costs.
__device__ __forceinline__ unsigned patch_claim_tile (const SkewLoopArgs & A)
{
  if (A.per_xcd == 0)
    return A.order[atomicAdd (A.ticket, 1u) + 1u];
  const unsigned x = patch_xcc_id ();
  unsigned tile = 0;
  for (unsigned q = 0; q < 8; q++) {
    const unsigned xx = (x + q) & 7;
    const unsigned k = atomicAdd (&A.xticket[xx], 1u) + 1u;
    if (k < (unsigned) A.per_xcd) {
      tile = A.xorder[xx*A.per_xcd + k];
      break;
    }
  }
  __hip_atomic_store (&A.tile_xcd[tile], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return tile;
}

// does tile `t' run on the XCD of this workgroup?  (waits for the claim of t: all tiles are resident)
__device__ __forceinline__ bool patch_same_xcd (const SkewLoopArgs & A, int t)
{
  if (A.per_xcd == 0) return false;
  const unsigned x = patch_xcc_id ();
  for (unsigned spins = 0; spins < (1u << 20); spins++) {
    const unsigned v = __hip_atomic_load (&A.tile_xcd[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != ~0u) return v == x;
    __builtin_amdgcn_s_sleep (4);
  }
  return false;       /* never claimed: the hand-off waits will report it */
}

// one cell: relax, src/poisson.c:507-530, unit weights, d = 0..5 = right, left, top, bottom, front, back.
// Without dia the quotient by 6 comes from divide_by_6's reciprocal sequence (relax_skew.hpp), whose
// guard -- operands so small that the sequence could meet subnormals -- is only recorded here: the
// four cells of a step form one dependent chain, a branch per cell would cut it into basic blocks
// that the scheduler cannot interleave.  The step tests the four flags once and, in the (never
// observed) case that one is set, recomputes its cells with true divisions.
template <bool HAS_DIA, int OP = 0>
__device__ __forceinline__ double patch_cell (double right, double left, double top, double bottom,
					      double front, double back, double rhs, double dia,
					      bool & tiny, bool exact, double w = 1., double h2 = 1.)
{
  if (OP == 1)
    return diffusion_cell (right, left, top, bottom, front, back, rhs, dia, w, h2);
  double aa = HAS_DIA ? dia : 0., bb = 0.;
  aa += 1.; bb += 1.*right;
  aa += 1.; bb += 1.*left;
  aa += 1.; bb += 1.*top;
  aa += 1.; bb += 1.*bottom;
  aa += 1.; bb += 1.*front;
  aa += 1.; bb += 1.*back;
  if (HAS_DIA)
    return aa != 0. ? (bb - rhs)/aa : 0.;
  const double x = bb - rhs;
  if (exact)
    return x/6.;
  const double r = 0x1.5555555555555p-3;
  const double q = x*r;
  const double rem = __builtin_fma (- q, 6., x);
  tiny = tiny || __builtin_amdgcn_frexp_exp (x) < -999;
  return __builtin_fma (rem, r, q);
}

template <bool HAS_DIA>
__global__ void __launch_bounds__(PK_NTHREADS)
relax_patch_loop_kernel (SkewLoopArgs A)
{
  constexpr int XS = SK_T + 1;
  __shared__ double X[2][XS*XS];
  __shared__ double Y[2][XS*XS];
  __shared__ unsigned s_tile;

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane (tid0 >> 6);
  const int lane = tid0 & 63;
  const bool compute = wave == 0, loader = wave == 1, storer = wave == 2;
  const int n = A.L.n;
  const int ntj = A.ntj;
  const long tstride = (long) (A.RT + 2*SK_FP)*SK_NL;
  const long hstride = (long) SK_HROWS (n)*SK_T;

  if (tid0 == 0)
    s_tile = patch_claim_tile (A);
  __syncthreads ();
  const int tile = s_tile;
  const int P = tile % ntj, Q = tile / ntj;
  patch_arm_other_set (A, tile, tid0, blockDim.x);
  // periodic neighbours of the tile
  const int tJm = (P > 0 ? P - 1 : ntj - 1) + ntj*Q, tJp = (P + 1 < ntj ? P + 1 : 0) + ntj*Q;
  const int tKm = P + ntj*(Q > 0 ? Q - 1 : ntj - 1), tKp = P + ntj*(Q + 1 < ntj ? Q + 1 : 0);

  double * const ut = A.us + tile*tstride + SK_FP*SK_NL;
  const double * const rt = A.rs + tile*tstride + SK_FP*SK_NL;
  const double * const dt_ = HAS_DIA ? A.ds + tile*tstride + SK_FP*SK_NL : nullptr;

  // compute lane (PA, PB); helper lanes: strip / line g = 0..3, position m = 0..15
  const int PA = lane & 7, PB = lane >> 3;
  const int s = PA + PB;
  const int g = (lane >> 4) & 3, m = lane & 15, mh = m >> 1;
  const int jm = n - (SK_T*P + m), km = n - (SK_T*Q + m);
  bool failed = false;

  // steps of a sweep: the lanes are active at t = s .. s + n - 1, the store wave one step later
  const int T = (n + PK_SKEW + 1 + PK_D)/PK_D*PK_D;

  // LDS indices of the compute lane (line coordinates: X at (a + 1) + XS (b + 1), Y at a + XS b)
  const int a0 = 2*PA, b0 = 2*PB;
  const int xT = a0 + XS*(b0 + 1);           // new (a0 - 1, b0); + XS: (a0 - 1, b0 + 1)
  const int xF = (a0 + 1) + XS*b0;           // new (a0, b0 - 1); + 1: (a0 + 1, b0 - 1)
  const int yBo = (a0 + 2) + XS*b0;          // old (a0 + 2, b0); + XS: (a0 + 2, b0 + 1)
  const int yBk = a0 + XS*(b0 + 2);          // old (a0, b0 + 2); + 1: (a0 + 1, b0 + 2)
  const int xOwn = (a0 + 1) + XS*(b0 + 1);   // own new values: + 0, + 1, + XS, + XS + 1
  const int yOwn = a0 + XS*b0;               // own old values one cell ahead

  // natural coordinates of the four lines of the lane: p = da + 2 db
  int jl[4], kl[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    jl[p] = n - (SK_T*P + a0 + (p & 1));
    kl[p] = n - (SK_T*Q + b0 + (p >> 1));
  }
  // ghost cells at the two ends of the lines: natural ghosts for the first sweep, then the periodic
  // images (or the homogeneous BC of the side) kept in registers
  double ghostL[4] = { 0., 0., 0., 0. }, ghostR[4] = { 0., 0., 0., 0. };
  if (compute) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ghostL[p] = A.un[A.L.idx (0, jl[p], kl[p])];
      ghostR[p] = A.un[A.L.idx (n + 1, jl[p], kl[p])];
    }
  }

  for (int sw = 0; sw < A.nsweeps; sw++) {
    const bool more = sw + 1 < A.nsweeps;
    const bool write_ghosts = sw + 2 == A.nsweeps;
    u64 * const hbJ = A.hb + sw*A.hb_sweep, * const hbK = hbJ + A.hb_words;
    u64 * const snJ = hbK + A.hb_words, * const snK = snJ + A.hb_words;
    const u64 * const hbJp = hbJ - A.hb_sweep, * const hbKp = hbK - A.hb_sweep;   // previous sweep
    const u64 * const snJp = snJ - A.hb_sweep, * const snKp = snK - A.hb_sweep;

    if (A.stats && tid0 == 0)
      A.stats[2*(tile*SK_MAXF + sw)] = __builtin_amdgcn_s_memrealtime ();

    __syncthreads ();      // the LDS grids of the previous sweep are no longer read
    for (int q = tid0; q < 2*XS*XS; q += PK_NTHREADS) {
      (&X[0][0])[q] = 0.;
      (&Y[0][0])[q] = 0.;
    }
    __syncthreads ();

    if (loader) {
      // =========================== halo wave ===========================
      const u64 * qH = A.dummy;
      int hs = 0;
      bool handoff = false;       // sentinel-guarded granule stream
      double hsgn = 1.;           // sign of a homogeneous BC whose ghost this stream carries
      int xy_halo = 0;
      switch (g) {
      case 0: // new values of line (-1, m), row t
	if (P > 0)       { qH = hbJ + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[2] == 0.) { qH = hbJp + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snJp + (long) tile*hstride + m + (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[2]; }      /* own line a = 0: snapshot row I + mh + 7 = t + 7 */
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - mh, n + 1, km)); hs = 1; }
	xy_halo = 0 + XS*(m + 1);
	break;
      case 1: // new values of line (m, -1)
	if (Q > 0)       { qH = hbK + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[4] == 0.) { qH = hbKp + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snKp + (long) tile*hstride + m + (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[4]; }      /* own line b = 0, row t + 7 */
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - mh, jm, n + 1)); hs = 1; }
	xy_halo = (m + 1) + XS*0;
	break;
      case 2: // old values of line (16, m) = line (0, m) of tile (P + 1, Q) at I = t - 7 - mh
	if (sw > 0 && (P + 1 < ntj || A.sgn[3] == 0.)) { qH = snJp + (long) tJp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbJp + (long) tile*hstride + m - (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[3]; }      /* own line a = 15: hand-off row I + mh = t - 7 */
	else if (P + 1 < ntj) {
	  // element (I, 0, m) of the next tile: row I + mh = t - 7, column 128 (m & 1) + 16 mh
	  qH = (const u64 *) (A.us + tJp*tstride + SK_FP*SK_NL - (long) 7*SK_NL + 128*(m & 1) + 16*mh);
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (7 + mh), 0, km)); hs = 1; }
	xy_halo = SK_T + XS*m;
	break;
      default: // old values of line (m, 16) = line (m, 0) of tile (P, Q + 1) at I = t - 7 - mh
	if (sw > 0 && (Q + 1 < ntj || A.sgn[5] == 0.)) { qH = snKp + (long) tKp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbKp + (long) tile*hstride + m - (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[5]; }      /* own line b = 15 */
	else if (Q + 1 < ntj) {
	  // element (I, m, 0) of the tile behind: row t - 7, column 2 mh + (m & 1)
	  qH = (const u64 *) (A.us + tKp*tstride + SK_FP*SK_NL - (long) 7*SK_NL + 2*mh + (m & 1));
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (7 + mh), jm, 0)); hs = 1; }
	xy_halo = m + XS*SK_T;
      }
      double * const halo_dst0 = (g < 2 ? &X[0][0] : &Y[0][0]) + xy_halo;
      const u64 * const qH0 = qH;
      // a granule stream is awaited only at the steps at which its consumer lane is active
      const int hlag = g < 2 ? mh : 7 + mh;
      double pH[PK_DH];

#define PK_HALO(t_, q_, refill_)					\
      do {								\
	double hv = pH[q_];						\
	bool w = handoff && !failed && (unsigned) ((t_) - hlag) < (unsigned) n && \
	  (u64) __double_as_longlong (hv) == SK_SENTINEL;		\
	if (__builtin_expect (__any (w), 0)) {				\
	  unsigned spins = 0;						\
	  _Pragma ("nounroll")						\
	  while (__any (w)) {						\
	    __builtin_amdgcn_s_sleep (SK_POLL_SLEEP);			\
	    if (w) {							\
	      hv = __longlong_as_double ((long long) load_sc1 (qH0 + (long) (t_)*hs)); \
	      w = (u64) __double_as_longlong (hv) == SK_SENTINEL;	\
	    }								\
	    if (++spins > (1u << 18)) { *A.err = 1; failed = true; break; } \
	  }								\
	}								\
	halo_dst0[((t_) & 1)*(XS*XS)] = hv*hsgn;			\
	if (refill_) {							\
	  pH[q_] = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs; \
	}								\
      } while (0)

      // strip values of step 0, then the ring holds steps 1 .. PK_DH
      {
	pH[0] = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs;
	PK_HALO (0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < PK_DH; q++) {
	pH[q] = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs;
      }
      __syncthreads ();
      // the value of step t + 1 is put into LDS during step t: slot (t mod PK_DH) holds step t + 1
      for (int t0 = 0; t0 < T; t0 += PK_D) {
#pragma unroll
	for (int q = 0; q < PK_D; q++) {
	  const int t = t0 + q;
	  PK_HALO (t + 1, q % PK_DH, 1);
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
#undef PK_HALO
    }
    else if (storer) {
      // =========================== store wave ===========================
      // lane (g, m) owns one outgoing line; the value the compute wave produced at step t - 1 sits
      // in X[t & 1]
      //   g = 0  line (15, m)  hand-off to tile (P + 1, Q), also its periodic image when a sweep follows
      //   g = 1  line (m, 15)  hand-off to tile (P, Q + 1)
      //   g = 2  line (0, m)   snapshot for the next sweep of tile (P - 1, Q)
      //   g = 3  line (m, 0)   snapshot for the next sweep of tile (P, Q - 1)
      u64 * pS = (u64 *) A.dummy;
      bool sOn = false;
      int sX = 0, sLag = 0;       // LDS index of the line's new value; I = t - 1 - sLag
      double * pNat = nullptr;    // mirror mode: the line lies along a box side: natural address of cell i = t - sLag
      switch (g) {
      case 0: sOn = P + 1 < ntj || more; pS = hbJ + (long) tile*hstride + m - (long) 8*SK_T;   /* row t - 8 */
	sX = SK_T + XS*(m + 1); sLag = 7 + mh;
	if (A.mirror && P == ntj - 1) pNat = A.un + A.L.idx (0, 1, km) - sLag;
	break;
      case 1: sOn = Q + 1 < ntj || more; pS = hbK + (long) tile*hstride + m - (long) 8*SK_T;
	sX = (m + 1) + XS*SK_T; sLag = 7 + mh;
	if (A.mirror && Q == ntj - 1) pNat = A.un + A.L.idx (0, jm, 1) - sLag;
	break;
      case 2: sOn = more; pS = snJ + (long) tile*hstride + m + (long) 6*SK_T;                   /* row t + 6 */
	sX = 1 + XS*(m + 1); sLag = mh;
	if (A.mirror && P == 0) pNat = A.un + A.L.idx (0, n, km) - sLag;
	break;
      default: sOn = more; pS = snK + (long) tile*hstride + m + (long) 6*SK_T;
	sX = (m + 1) + XS*1; sLag = mh;
	if (A.mirror && Q == 0) pNat = A.un + A.L.idx (0, jm, n) - sLag;
      }
      // ... and every lane writes the row of the skewed copy that the compute lane of the same number
      // produced in the previous step (its four new values are in X): the stores of the rows stay out
      // of the compute wave's in-order memory queue, whose prefetched loads would wait behind them
      double * wRow = ut + 2*lane - SK_NL;            // row t - 1
      __syncthreads ();
      for (int t0 = 0; t0 < T; t0 += PK_D) {
#pragma unroll
	for (int q = 0; q < PK_D; q++) {
	  const int t = t0 + q;
	  const int I = t - 1 - sLag;
	  const double * Xb = X[t & 1];
	  if (!(PK_KO & 2) && t > 0) {
	    d2 o01, o23;
	    o01.x = Xb[xOwn]; o01.y = Xb[xOwn + 1]; o23.x = Xb[xOwn + XS]; o23.y = Xb[xOwn + XS + 1];
	    *(d2 *) wRow = o01; *(d2 *) (wRow + 128) = o23;
	  }
	  wRow += SK_NL;
	  if (I >= 0 && I < n) {
	    const double v = Xb[sX];
	    if (pNat) pNat[t] = v;
	    if (sOn) store_sc1 (pS, (u64) __double_as_longlong (v));
	  }
	  pS += SK_T;
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
    }
    else {
      // =========================== compute wave ===========================
      const double * qR = ut + SK_NL + 2*lane;        // old values one cell ahead: row t + 1
      const double * qRhs = rt + 2*lane;              // row t
      const double * qDia = HAS_DIA ? dt_ + 2*lane : nullptr;
      d2 pRa[PK_D], pRb[PK_D], pHa[PK_D], pHb[PK_D], pDa[PK_D], pDb[PK_D];
#define PK_PREFETCH(q_)							\
      do {								\
	pRa[q_] = *(const d2 *) qR; pRb[q_] = *(const d2 *) (qR + 128); qR += SK_NL; \
	pHa[q_] = *(const d2 *) qRhs; pHb[q_] = *(const d2 *) (qRhs + 128); qRhs += SK_NL; \
	if (HAS_DIA) { pDa[q_] = *(const d2 *) qDia; pDb[q_] = *(const d2 *) (qDia + 128); qDia += SK_NL; } \
      } while (0)
      // old values of the own cells at I = 0 - s ... : row t of the first step
      d2 c01 = *(const d2 *) (ut + 2*lane), c23 = *(const d2 *) (ut + 2*lane + 128);
#pragma unroll
      for (int q = 0; q < PK_D; q++)
	PK_PREFETCH (q);
      // the old values one cell ahead that the neighbours read at step 0
      Y[0][yOwn] = pRa[0].x; Y[0][yOwn + 1] = pRa[0].y;
      Y[0][yOwn + XS] = pRb[0].x; Y[0][yOwn + XS + 1] = pRb[0].y;
      __syncthreads ();

      double prev[4] = { ghostL[0], ghostL[1], ghostL[2], ghostL[3] };
      double first[4] = { 0., 0., 0., 0. }, lastv[4] = { 0., 0., 0., 0. };
      double cur[4] = { c01.x, c01.y, c23.x, c23.y };

      for (int t0 = 0; t0 < T; t0 += PK_D) {
#pragma unroll
	for (int q = 0; q < PK_D; q++) {
	  const int t = t0 + q;
	  const int I = t - s;
	  const int Bf = t & 1;
	  const double * Xb = X[Bf], * Yb = Y[Bf];
	  const double Tn0 = Xb[xT], Tn1 = Xb[xT + XS];
	  const double Fn0 = Xb[xF], Fn1 = Xb[xF + 1];
	  const double Bo0 = Yb[yBo], Bo1 = Yb[yBo + XS];
	  const double Bk0 = Yb[yBk], Bk1 = Yb[yBk + 1];
	  double nx[4] = { pRa[q].x, pRa[q].y, pRb[q].x, pRb[q].y };   // old values at I + 1
	  const double keep[4] = { nx[0], nx[1], nx[2], nx[3] };
	  // the two ends of the lines (one step each per lane and sweep: a rarely entered block instead
	  // of selects in every step): left ghost at I = 0, right ghost at I = n - 1
	  if (__builtin_expect (I == 0 || I == n - 1, 0)) {
#pragma unroll
	    for (int p = 0; p < 4; p++) {
	      prev[p] = I == 0 ? ghostL[p] : prev[p];
	      nx[p] = I == n - 1 ? ghostR[p] : nx[p];
	    }
	  }
	  const double d0 = HAS_DIA ? pDa[q].x : 0., d1 = HAS_DIA ? pDa[q].y : 0.;
	  const double d2_ = HAS_DIA ? pDb[q].x : 0., d3 = HAS_DIA ? pDb[q].y : 0.;
	  bool tiny = false;
	  // (a0, b0): top and front from the neighbour lanes, bottom and back the lane's own old values
	  double v0 = patch_cell<HAS_DIA> (nx[0], prev[0], Tn0, cur[1], Fn0, cur[2], pHa[q].x, d0, tiny, false);
	  // (a0 + 1, b0): top = the new (a0, b0)
	  double v1 = patch_cell<HAS_DIA> (nx[1], prev[1], v0, Bo0, Fn1, cur[3], pHa[q].y, d1, tiny, false);
	  // (a0, b0 + 1): front = the new (a0, b0)
	  double v2 = patch_cell<HAS_DIA> (nx[2], prev[2], Tn1, cur[3], v0, Bk0, pHb[q].x, d2_, tiny, false);
	  // (a0 + 1, b0 + 1)
	  double v3 = patch_cell<HAS_DIA> (nx[3], prev[3], v2, Bo1, v1, Bk1, pHb[q].y, d3, tiny, false);
	  if (!HAS_DIA && __builtin_expect (__builtin_amdgcn_ballot_w64 (tiny) != 0, 0)) {
	    v0 = patch_cell<HAS_DIA> (nx[0], prev[0], Tn0, cur[1], Fn0, cur[2], pHa[q].x, d0, tiny, true);
	    v1 = patch_cell<HAS_DIA> (nx[1], prev[1], v0, Bo0, Fn1, cur[3], pHa[q].y, d1, tiny, true);
	    v2 = patch_cell<HAS_DIA> (nx[2], prev[2], Tn1, cur[3], v0, Bk0, pHb[q].x, d2_, tiny, true);
	    v3 = patch_cell<HAS_DIA> (nx[3], prev[3], v2, Bo1, v1, Bk1, pHb[q].y, d3, tiny, true);
	  }
	  prev[0] = v0; prev[1] = v1; prev[2] = v2; prev[3] = v3;
	  if (__builtin_expect (I == 0 || I == n - 1, 0)) {
	    const double vv[4] = { v0, v1, v2, v3 };
#pragma unroll
	    for (int p = 0; p < 4; p++) {
	      first[p] = I == 0 ? vv[p] : first[p];
	      lastv[p] = I == n - 1 ? vv[p] : lastv[p];
	    }
	  }
	  // publish for step t + 1: new values, and the old values one cell ahead of step t + 1
	  double * Xn = X[Bf ^ 1], * Yn = Y[Bf ^ 1];
	  Xn[xOwn] = v0; Xn[xOwn + 1] = v1; Xn[xOwn + XS] = v2; Xn[xOwn + XS + 1] = v3;
	  {
	    const int qn = (q + 1) % PK_D;
	    Yn[yOwn] = pRa[qn].x; Yn[yOwn + 1] = pRa[qn].y;
	    Yn[yOwn + XS] = pRb[qn].x; Yn[yOwn + XS + 1] = pRb[qn].y;
	  }
	  cur[0] = keep[0]; cur[1] = keep[1]; cur[2] = keep[2]; cur[3] = keep[3];
	  if (!(PK_KO & 1))
	    PK_PREFETCH (q);
	  // (the own row of the skewed copy is written by the store wave, from X, one step later)
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
#undef PK_PREFETCH
      // ghosts of the lines for the next sweep: the periodic image, or the homogeneous BC of the side
#pragma unroll
      for (int p = 0; p < 4; p++) {
	ghostL[p] = A.sgn[1] == 0. ? lastv[p] : A.sgn[1]*first[p];     // I = -1 (left side)
	ghostR[p] = A.sgn[0] == 0. ? first[p] : A.sgn[0]*lastv[p];     // I = n  (right side)
      }
      if (A.mirror) {
#pragma unroll
	for (int p = 0; p < 4; p++) {
	  A.un[A.L.idx (1, jl[p], kl[p])] = first[p];
	  A.un[A.L.idx (n, jl[p], kl[p])] = lastv[p];
	}
      }
      if (write_ghosts) {
	// x ghosts of the last BC application (the y and z ghost planes: patch_loop_ghosts_kernel)
#pragma unroll
	for (int p = 0; p < 4; p++) {
	  A.un[A.L.idx (n + 1, jl[p], kl[p])] = ghostR[p];
	  A.un[A.L.idx (0, jl[p], kl[p])] = ghostL[p];
	}
      }
    }
    if (A.stats && tid0 == 0)
      A.stats[2*(tile*SK_MAXF + sw) + 1] = __builtin_amdgcn_s_memrealtime ();
  }
}

// ---------------------------------------------------------------------------------------------
// The same with the rows streamed through LDS by a fourth wave (LDS-DMA), so that the compute wave
// issues no vector-memory instruction at all: with one wave computing the tile the step is bound by
// what that wave issues (4 cycles per instruction), and the four 16-byte loads of a step, their
// address arithmetic and the waits for them cost a quarter of it (PK_KO experiments: 63 us per
// tile-sweep without them, 85 us with them at 256^3).  The stream wave puts row t + 1 + RK_DS of u and
// row t + RK_DS of rhs (dia) into a ring of RK_R slots in LDS at step t (global_load_lds, 1 KB per
// instruction, no registers), retiring them with a counted s_waitcnt one step before they are read.
// The compute lane reads its own cells of row t + 1 from the ring -- and the old values of the
// lines after its patch from the same row (they are its neighbour lanes' cells): the Y grid and
// its four LDS writes per step are gone; the `old' strips of the halo wave sit next to the row in
// the slot.  Everything else (X grid, store wave, granules, layout) is as above.
// ---------------------------------------------------------------------------------------------
// Waiting for a hand-off granule with TWO polls in flight, half a round trip apart: a poll that is
// only re-issued when the previous one has come back notices the granule on average half a round
// trip (0.7 - 1 us under load) after it has become visible; with two staggered polls that is a
// quarter.  Agent-scope loads by inline assembly (the compiler would wait for each before issuing
// the next); the "+v" operands of the waits tie the uses of the values to them.  w: lanes that wait
// (the others keep hv); every lane leaves together (wave-uniform loop).
#ifndef RK_POLL2
#define RK_POLL2 0     /* 1: measured 0.540 against 0.546 ms per loop: within the noise of the box */
#endif
__device__ __forceinline__ double poll2 (const u64 * p, double hv, bool w, unsigned * err, bool & failed)
{
  u64 va = SK_SENTINEL, vb = SK_SENTINEL;
  const u64 * q = w ? p : nullptr;
  unsigned spins = 0;
  if (w) asm volatile ("global_load_dwordx2 %0, %1, off sc1" : "=v" (va) : "v" (q) : "memory");
  __builtin_amdgcn_s_sleep (6);        /* the stagger: ~ 0.2 us */
  if (w) asm volatile ("global_load_dwordx2 %0, %1, off sc1" : "=v" (vb) : "v" (q) : "memory");
  for (;;) {
    asm volatile ("s_waitcnt vmcnt(1)" : "+v" (va) :: "memory");
    if (w && va != SK_SENTINEL) { hv = __longlong_as_double ((long long) va); w = false; }
    if (!__any (w)) break;
    if (w) asm volatile ("global_load_dwordx2 %0, %1, off sc1" : "=v" (va) : "v" (q) : "memory");
    asm volatile ("s_waitcnt vmcnt(1)" : "+v" (vb) :: "memory");
    if (w && vb != SK_SENTINEL) { hv = __longlong_as_double ((long long) vb); w = false; }
    if (!__any (w)) break;
    if (w) asm volatile ("global_load_dwordx2 %0, %1, off sc1" : "=v" (vb) : "v" (q) : "memory");
    if (++spins > (1u << 17)) { *err = 1; failed = true; break; }
  }
  asm volatile ("s_waitcnt vmcnt(0)" : "+v" (va), "+v" (vb) :: "memory");
  return hv;
}

#ifndef RK_R
#define RK_R 12      /* ring slots = unroll factor of the step loop (slot numbers are compile-time) */
#endif
#ifndef RK_DS
#define RK_DS 9      /* steps the stream wave runs ahead (<= RK_R - 2) */
#endif
#define RK_NTHREADS 256
#ifndef RK_PAIR_STORES
#define RK_PAIR_STORES 0   /* 1: hand-off and snapshot granules of two lines in one 16-byte sc1 store (bit-identical; measured 2 % slower: 0.557 against 0.547 ms per loop) */
#endif

template <bool HAS_DIA, int OP>
__global__ void __launch_bounds__(RK_NTHREADS)
relax_ring_loop_kernel (SkewLoopArgs A)
{
  constexpr int XS = SK_T + 1;
  // doubles per ring slot: the u row, the rhs row (the dia row), and the two `old' halo strips
  constexpr int SS = (HAS_DIA ? 3 : 2)*SK_NL + 32;
  constexpr int oRhs = SK_NL, oDia = 2*SK_NL, oHalo = (HAS_DIA ? 3 : 2)*SK_NL;
  __shared__ double X[2][XS*XS];
  __shared__ __attribute__((aligned(16))) double ring[RK_R*SS];
  __shared__ unsigned s_tile;

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane (tid0 >> 6);
  const int lane = tid0 & 63;
  const bool compute = wave == 0, loader = wave == 1, storer = wave == 2;   // wave 3: the stream wave
  const int n = A.L.n;
  const int ntj = A.ntj;
  const long tstride = (long) (A.RT + 2*SK_FP)*SK_NL;
  const long hstride = (long) SK_HROWS (n)*SK_T;

  if (tid0 == 0)
    s_tile = patch_claim_tile (A);
  __syncthreads ();
  const int tile = s_tile;
  const int P = tile % ntj, Q = tile / ntj;
  patch_arm_other_set (A, tile, tid0, blockDim.x);
  // periodic neighbours of the tile
  const int tJm = (P > 0 ? P - 1 : ntj - 1) + ntj*Q, tJp = (P + 1 < ntj ? P + 1 : 0) + ntj*Q;
  const int tKm = P + ntj*(Q > 0 ? Q - 1 : ntj - 1), tKp = P + ntj*(Q + 1 < ntj ? Q + 1 : 0);

  double * const ut = A.us + tile*tstride + SK_FP*SK_NL;
  const double * const rt = A.rs + tile*tstride + SK_FP*SK_NL;
  const double * const dt_ = HAS_DIA ? A.ds + tile*tstride + SK_FP*SK_NL : nullptr;

  // compute lane (PA, PB); helper lanes: strip / line g = 0..3, position m = 0..15
  const int PA = lane & 7, PB = lane >> 3;
  const int s = PA + PB;
  const int g = (lane >> 4) & 3, m = lane & 15, mh = m >> 1;
  const int jm = n - (SK_T*P + m), km = n - (SK_T*Q + m);
  bool failed = false;

  // steps of a sweep: the lanes are active at t = s .. s + n - 1, the store wave one step later
  const int T = (n + PK_SKEW + 1 + RK_R)/RK_R*RK_R;

  // LDS indices of the compute lane (line coordinates: X at (a + 1) + XS (b + 1), Y at a + XS b)
  const int a0 = 2*PA, b0 = 2*PB;
  const int xT = a0 + XS*(b0 + 1);           // new (a0 - 1, b0); + XS: (a0 - 1, b0 + 1)
  const int xF = (a0 + 1) + XS*b0;           // new (a0, b0 - 1); + 1: (a0 + 1, b0 - 1)
  const int xOwn = (a0 + 1) + XS*(b0 + 1);   // own new values: + 0, + 1, + XS, + XS + 1
  // the old values at I of the lines after the patch are elements of row t + 1 of the lanes PA + 1 /
  // PB + 1 -- in the ring, where the stream wave has put that row -- or, along the + sides of the
  // tile, the strip values the halo wave puts next to it (slot offsets in doubles):
  //   (a0 + 2, b0), (a0 + 2, b0 + 1): lane + 1, its cells p = 0 (first half) and p = 2 (second half)
  //   (a0, b0 + 2), (a0 + 1, b0 + 2): lane + 8, its cells p = 0, 1 (16 contiguous bytes)
  const int rOwn = 2*lane;
  const int rBo0 = PA < 7 ? 2*(lane + 1) : oHalo + b0;
  const int rBo1 = PA < 7 ? 128 + 2*(lane + 1) : oHalo + b0 + 1;
  const int rBk = PB < 7 ? 2*(lane + 8) : oHalo + 16 + a0;

  // natural coordinates of the four lines of the lane: p = da + 2 db
  int jl[4], kl[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    jl[p] = n - (SK_T*P + a0 + (p & 1));
    kl[p] = n - (SK_T*Q + b0 + (p >> 1));
  }
  // ghost cells at the two ends of the lines: natural ghosts for the first sweep, then the periodic
  // images (or the homogeneous BC of the side) kept in registers
  double ghostL[4] = { 0., 0., 0., 0. }, ghostR[4] = { 0., 0., 0., 0. };
  if (compute) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ghostL[p] = A.un[A.L.idx (0, jl[p], kl[p])];
      ghostR[p] = A.un[A.L.idx (n + 1, jl[p], kl[p])];
    }
  }

  for (int sw = 0; sw < A.nsweeps; sw++) {
    const bool more = sw + 1 < A.nsweeps;
    const bool write_ghosts = sw + 2 == A.nsweeps;
    u64 * const hbJ = A.hb + sw*A.hb_sweep, * const hbK = hbJ + A.hb_words;
    u64 * const snJ = hbK + A.hb_words, * const snK = snJ + A.hb_words;
    const u64 * const hbJp = hbJ - A.hb_sweep, * const hbKp = hbK - A.hb_sweep;   // previous sweep
    const u64 * const snJp = snJ - A.hb_sweep, * const snKp = snK - A.hb_sweep;

    if (A.stats && tid0 == 0)
      A.stats[2*(tile*SK_MAXF + sw)] = __builtin_amdgcn_s_memrealtime ();

    __syncthreads ();      // the LDS grids of the previous sweep are no longer read
    for (int q = tid0; q < 2*XS*XS; q += RK_NTHREADS)
      (&X[0][0])[q] = 0.;
    __syncthreads ();

    if (loader) {
      // =========================== halo wave ===========================
      const u64 * qH = A.dummy;
      int hs = 0;
      bool handoff = false;       // sentinel-guarded granule stream
      double hsgn = 1.;           // sign of a homogeneous BC whose ghost this stream carries
      int xy_halo = 0;
      switch (g) {
      case 0: // new values of line (-1, m), row t
	if (P > 0)       { qH = hbJ + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[2] == 0.) { qH = hbJp + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snJp + (long) tile*hstride + m + (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[2]; }      /* own line a = 0: snapshot row I + mh + 7 = t + 7 */
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - mh, n + 1, km)); hs = 1; }
	xy_halo = 0 + XS*(m + 1);
	break;
      case 1: // new values of line (m, -1)
	if (Q > 0)       { qH = hbK + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[4] == 0.) { qH = hbKp + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snKp + (long) tile*hstride + m + (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[4]; }      /* own line b = 0, row t + 7 */
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - mh, jm, n + 1)); hs = 1; }
	xy_halo = (m + 1) + XS*0;
	break;
      case 2: // old values of line (16, m) = line (0, m) of tile (P + 1, Q) at I = t - 7 - mh
	if (sw > 0 && (P + 1 < ntj || A.sgn[3] == 0.)) { qH = snJp + (long) tJp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbJp + (long) tile*hstride + m - (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[3]; }      /* own line a = 15: hand-off row I + mh = t - 7 */
	else if (P + 1 < ntj) {
	  // element (I, 0, m) of the next tile: row I + mh = t - 7, column 128 (m & 1) + 16 mh
	  qH = (const u64 *) (A.us + tJp*tstride + SK_FP*SK_NL - (long) 7*SK_NL + 128*(m & 1) + 16*mh);
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (7 + mh), 0, km)); hs = 1; }
	xy_halo = SK_T + XS*m;
	break;
      default: // old values of line (m, 16) = line (m, 0) of tile (P, Q + 1) at I = t - 7 - mh
	if (sw > 0 && (Q + 1 < ntj || A.sgn[5] == 0.)) { qH = snKp + (long) tKp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbKp + (long) tile*hstride + m - (long) 7*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[5]; }      /* own line b = 15 */
	else if (Q + 1 < ntj) {
	  // element (I, m, 0) of the tile behind: row t - 7, column 2 mh + (m & 1)
	  qH = (const u64 *) (A.us + tKp*tstride + SK_FP*SK_NL - (long) 7*SK_NL + 2*mh + (m & 1));
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (7 + mh), jm, 0)); hs = 1; }
	xy_halo = m + XS*SK_T;
      }
      // new values go into the X grid of the step's parity; old values into the slot of row t + 1
      double * const halo_x = &X[0][0] + xy_halo;
      double * const halo_y = ring + oHalo + (g == 2 ? m : 16 + m);
      const u64 * const qH0 = qH;
      // a granule stream is awaited only at the steps at which its consumer lane is active
      const int hlag = g < 2 ? mh : 7 + mh;
      double pH[PK_DH];

#define PK_HALO(t_, q_, refill_)					\
      do {								\
	double hv = pH[q_];						\
	bool w = handoff && !failed && (unsigned) ((t_) - hlag) < (unsigned) n && \
	  (u64) __double_as_longlong (hv) == SK_SENTINEL;		\
	if (__builtin_expect (__any (w), 0)) {				\
	  if (RK_POLL2)							\
	    hv = poll2 (qH0 + (long) (t_)*hs, hv, w, A.err, failed);	\
	  else {							\
	  unsigned spins = 0;						\
	  _Pragma ("nounroll")						\
	  while (__any (w)) {						\
	    __builtin_amdgcn_s_sleep (SK_POLL_SLEEP);			\
	    if (w) {							\
	      hv = __longlong_as_double ((long long) load_sc1 (qH0 + (long) (t_)*hs)); \
	      w = (u64) __double_as_longlong (hv) == SK_SENTINEL;	\
	    }								\
	    if (++spins > (1u << 18)) { *A.err = 1; failed = true; break; } \
	  }								\
	  }								\
	}								\
	if (g < 2) halo_x[((t_) & 1)*(XS*XS)] = hv*hsgn;		\
	else halo_y[(((t_) + 1) % RK_R)*SS] = hv*hsgn;			\
	if (refill_) {							\
	  pH[q_] = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs; \
	}								\
      } while (0)

      // strip values of step 0, then the ring holds steps 1 .. PK_DH
      {
	pH[0] = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs;
	PK_HALO (0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < PK_DH; q++) {
	pH[q] = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs;
      }
      __syncthreads ();
      // the value of step t + 1 is put into LDS during step t: slot (t mod PK_DH) holds step t + 1
      for (int t0 = 0; t0 < T; t0 += RK_R) {
#pragma unroll
	for (int q = 0; q < RK_R; q++) {
	  const int t = t0 + q;
	  PK_HALO (t + 1, q % PK_DH, 1);
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
#undef PK_HALO
    }
    else if (storer) {
      // =========================== store wave ===========================
      // lane (g, m) owns one outgoing line; the value the compute wave produced at step t - 1 sits
      // in X[t & 1]
      //   g = 0  line (15, m)  hand-off to tile (P + 1, Q), also its periodic image when a sweep follows
      //   g = 1  line (m, 15)  hand-off to tile (P, Q + 1)
      //   g = 2  line (0, m)   snapshot for the next sweep of tile (P - 1, Q)
      //   g = 3  line (m, 0)   snapshot for the next sweep of tile (P, Q - 1)
      u64 * pS = (u64 *) A.dummy;
      bool sOn = false;
      int sX = 0, sLag = 0;       // LDS index of the line's new value; I = t - 1 - sLag
      double * pNat = nullptr;    // mirror mode: the line lies along a box side: natural address of cell i = t - sLag
      switch (g) {
      case 0: sOn = P + 1 < ntj || more; pS = hbJ + (long) tile*hstride + m - (long) 8*SK_T;   /* row t - 8 */
	sX = SK_T + XS*(m + 1); sLag = 7 + mh;
	if (A.mirror && P == ntj - 1) pNat = A.un + A.L.idx (0, 1, km) - sLag;
	break;
      case 1: sOn = Q + 1 < ntj || more; pS = hbK + (long) tile*hstride + m - (long) 8*SK_T;
	sX = (m + 1) + XS*SK_T; sLag = 7 + mh;
	if (A.mirror && Q == ntj - 1) pNat = A.un + A.L.idx (0, jm, 1) - sLag;
	break;
      case 2: sOn = more; pS = snJ + (long) tile*hstride + m + (long) 6*SK_T;                   /* row t + 6 */
	sX = 1 + XS*(m + 1); sLag = mh;
	if (A.mirror && P == 0) pNat = A.un + A.L.idx (0, n, km) - sLag;
	break;
      default: sOn = more; pS = snK + (long) tile*hstride + m + (long) 6*SK_T;
	sX = (m + 1) + XS*1; sLag = mh;
	if (A.mirror && Q == 0) pNat = A.un + A.L.idx (0, jm, n) - sLag;
      }
      // LDS index of the new value of the line m + 1 (the partner of an even m in a 16-byte store)
      const int sX2 = g == 0 ? SK_T + XS*(m + 2) : g == 1 ? (m + 2) + XS*SK_T : g == 2 ? 1 + XS*(m + 2) : (m + 2) + XS*1;
      (void) sX2;
      if (tile == A.fault_tile && sw == 0) sOn = false;     /* fault injection for the test of the error path */
      // the consumer tile of the stream runs on this XCD: plain stores (they stay in the common L2)
      const bool sNear = A.near_mode != 0 && patch_same_xcd (A, g == 0 ? tJp : g == 1 ? tKp : g == 2 ? tJm : tKm);
      // ... and every lane writes the row of the skewed copy that the compute lane of the same number
      // produced in the previous step (its four new values are in X): the stores of the rows stay out
      // of the compute wave's in-order memory queue, whose prefetched loads would wait behind them
      double * wRow = ut + 2*lane - SK_NL;            // row t - 1
      __syncthreads ();
      for (int t0 = 0; t0 < T; t0 += RK_R) {
#pragma unroll
	for (int q = 0; q < RK_R; q++) {
	  const int t = t0 + q;
	  const int I = t - 1 - sLag;
	  const double * Xb = X[t & 1];
	  if (!(PK_KO & 2) && t > 0) {
	    d2 o01, o23;
	    o01.x = Xb[xOwn]; o01.y = Xb[xOwn + 1]; o23.x = Xb[xOwn + XS]; o23.y = Xb[xOwn + XS + 1];
	    *(d2 *) wRow = o01; *(d2 *) (wRow + 128) = o23;
	  }
	  wRow += SK_NL;
	  if (I >= 0 && I < n) {
	    const double v = Xb[sX];
	    if (pNat) pNat[t] = v;
	    if (sOn) {
	      if (sNear) {
		if (A.near_mode == 1) *(volatile u64 *) pS = (u64) __double_as_longlong (v);
		else __hip_atomic_store ((gu64 *) pS, (u64) __double_as_longlong (v), __ATOMIC_RELAXED,
					 __HIP_MEMORY_SCOPE_WORKGROUP);
	      }
#if RK_PAIR_STORES
	      else if (!(m & 1)) {
		/* the granules of the lines m, m + 1 (same lag: I depends on m >> 1) as ONE 16-byte agent-scope
		   store: half as many fabric writes per step (every 8-byte sc1 store is a write of its own); the
		   consumer polls 8-byte halves, which are never torn */
		typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
		ull2 w;
		w.x = (u64) __double_as_longlong (v);
		w.y = (u64) __double_as_longlong (Xb[sX2]);
		asm volatile ("global_store_dwordx4 %0, %1, off sc1" :: "v" (pS), "v" (w) : "memory");
	      }
#else
	      else store_sc1 (pS, (u64) __double_as_longlong (v));
#endif
	    }
	  }
	  pS += SK_T;
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
    }
    else if (wave == 3) {
      // =========================== stream wave ===========================
      // LDS-DMA (global_load_lds, no register destination) of the rows into the ring: RK_DS steps
      // ahead, one KB per instruction.  At step t: the u row t + 1 + RK_DS and the rhs (dia) row
      // t + RK_DS are issued; the counted wait then retires what was issued RK_DS - 1 steps ago, i.e.
      // the rows of step t + 1, before the barrier in front of that step.
      const double * gU = ut + 2*lane, * gR = rt + 2*lane, * gD = HAS_DIA ? dt_ + 2*lane : nullptr;
      constexpr int NI = HAS_DIA ? 6 : 4;
#define RK_LOAD_ROW(src_, row_, dst_)						\
      do {									\
	__builtin_amdgcn_global_load_lds ((const void *) ((src_) + (long) (row_)*SK_NL),	\
					  (__attribute__((address_space(3))) void *) (dst_), 16, 0, 0); \
	__builtin_amdgcn_global_load_lds ((const void *) ((src_) + (long) (row_)*SK_NL + 128),	\
					  (__attribute__((address_space(3))) void *) ((dst_) + 128), 16, 0, 0); \
      } while (0)
      // prologue: u rows 0 .. RK_DS, rhs rows 0 .. RK_DS - 1
#pragma unroll
      for (int r = 0; r <= RK_DS; r++)
	RK_LOAD_ROW (gU, r, ring + (r % RK_R)*SS);
#pragma unroll
      for (int r = 0; r < RK_DS; r++) {
	RK_LOAD_ROW (gR, r, ring + (r % RK_R)*SS + oRhs);
	if (HAS_DIA) RK_LOAD_ROW (gD, r, ring + (r % RK_R)*SS + oDia);
      }
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads ();
      for (int t0 = 0; t0 < T; t0 += RK_R) {
#pragma unroll
	for (int q = 0; q < RK_R; q++) {
	  const int t = t0 + q;
	  if (!(PK_KO & 1)) {
	    RK_LOAD_ROW (gU, t + 1 + RK_DS, ring + ((q + 1 + RK_DS) % RK_R)*SS);
	    RK_LOAD_ROW (gR, t + RK_DS, ring + ((q + RK_DS) % RK_R)*SS + oRhs);
	    if (HAS_DIA) RK_LOAD_ROW (gD, t + RK_DS, ring + ((q + RK_DS) % RK_R)*SS + oDia);
	    {
	      // s_waitcnt vmcnt (NI*(RK_DS - 1)) alone: gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]
	      constexpr int N = NI*(RK_DS - 1);
	      static_assert (N < 64, "the vmcnt counter holds 63");
	      __builtin_amdgcn_s_waitcnt ((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
	    }
	  }
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
#undef RK_LOAD_ROW
    }
    else {
      // =========================== compute wave ===========================
      __syncthreads ();
      // old values of the own cells at the I of step 0: row 0
      const d2 c01 = *(const d2 *) (ring + rOwn), c23 = *(const d2 *) (ring + rOwn + 128);
      double prev[4] = { ghostL[0], ghostL[1], ghostL[2], ghostL[3] };
      double first[4] = { 0., 0., 0., 0. }, lastv[4] = { 0., 0., 0., 0. };
      double cur[4] = { c01.x, c01.y, c23.x, c23.y };

      for (int t0 = 0; t0 < T; t0 += RK_R) {
#pragma unroll
	for (int q = 0; q < RK_R; q++) {
	  const int t = t0 + q;
	  const int I = t - s;
	  const int Bf = t & 1;
	  const double * Xb = X[Bf];
	  const double * const su = ring + ((q + 1) % RK_R)*SS;      // slot of the u row t + 1
	  const double * const sr = ring + (q % RK_R)*SS;            // slot of the rhs (dia) row t
	  const double Tn0 = Xb[xT], Tn1 = Xb[xT + XS];
	  const double Fn0 = Xb[xF], Fn1 = Xb[xF + 1];
	  const double Bo0 = su[rBo0], Bo1 = su[rBo1];
	  const d2 bk = *(const d2 *) (su + rBk);
	  const double Bk0 = bk.x, Bk1 = bk.y;
	  const d2 n01 = *(const d2 *) (su + rOwn), n23 = *(const d2 *) (su + rOwn + 128);
	  const d2 h01 = *(const d2 *) (sr + oRhs + rOwn), h23 = *(const d2 *) (sr + oRhs + rOwn + 128);
	  d2 e01 = { 0., 0. }, e23 = { 0., 0. };
	  if (HAS_DIA) { e01 = *(const d2 *) (sr + oDia + rOwn); e23 = *(const d2 *) (sr + oDia + rOwn + 128); }
	  double nx[4] = { n01.x, n01.y, n23.x, n23.y };   // old values at I + 1
	  const double keep[4] = { nx[0], nx[1], nx[2], nx[3] };
	  // the two ends of the lines (one step each per lane and sweep: a rarely entered block instead
	  // of selects in every step): left ghost at I = 0, right ghost at I = n - 1
	  if (__builtin_expect (I == 0 || I == n - 1, 0)) {
#pragma unroll
	    for (int p = 0; p < 4; p++) {
	      prev[p] = I == 0 ? ghostL[p] : prev[p];
	      nx[p] = I == n - 1 ? ghostR[p] : nx[p];
	    }
	  }
	  bool tiny = false;
	  // (a0, b0): top and front from the neighbour lanes, bottom and back the lane's own old values
	  double v0 = patch_cell<HAS_DIA, OP> (nx[0], prev[0], Tn0, cur[1], Fn0, cur[2], h01.x, e01.x, tiny, false, A.w, A.h2);
	  // (a0 + 1, b0): top = the new (a0, b0)
	  double v1 = patch_cell<HAS_DIA, OP> (nx[1], prev[1], v0, Bo0, Fn1, cur[3], h01.y, e01.y, tiny, false, A.w, A.h2);
	  // (a0, b0 + 1): front = the new (a0, b0)
	  double v2 = patch_cell<HAS_DIA, OP> (nx[2], prev[2], Tn1, cur[3], v0, Bk0, h23.x, e23.x, tiny, false, A.w, A.h2);
	  // (a0 + 1, b0 + 1)
	  double v3 = patch_cell<HAS_DIA, OP> (nx[3], prev[3], v2, Bo1, v1, Bk1, h23.y, e23.y, tiny, false, A.w, A.h2);
	  if (!HAS_DIA && __builtin_expect (__builtin_amdgcn_ballot_w64 (tiny) != 0, 0)) {
	    v0 = patch_cell<HAS_DIA> (nx[0], prev[0], Tn0, cur[1], Fn0, cur[2], h01.x, e01.x, tiny, true);
	    v1 = patch_cell<HAS_DIA> (nx[1], prev[1], v0, Bo0, Fn1, cur[3], h01.y, e01.y, tiny, true);
	    v2 = patch_cell<HAS_DIA> (nx[2], prev[2], Tn1, cur[3], v0, Bk0, h23.x, e23.x, tiny, true);
	    v3 = patch_cell<HAS_DIA> (nx[3], prev[3], v2, Bo1, v1, Bk1, h23.y, e23.y, tiny, true);
	  }
	  prev[0] = v0; prev[1] = v1; prev[2] = v2; prev[3] = v3;
	  if (__builtin_expect (I == 0 || I == n - 1, 0)) {
	    const double vv[4] = { v0, v1, v2, v3 };
#pragma unroll
	    for (int p = 0; p < 4; p++) {
	      first[p] = I == 0 ? vv[p] : first[p];
	      lastv[p] = I == n - 1 ? vv[p] : lastv[p];
	    }
	  }
	  // publish the new values for step t + 1 (the row itself is written by the store wave)
	  double * Xn = X[Bf ^ 1];
	  Xn[xOwn] = v0; Xn[xOwn + 1] = v1; Xn[xOwn + XS] = v2; Xn[xOwn + XS + 1] = v3;
	  cur[0] = keep[0]; cur[1] = keep[1]; cur[2] = keep[2]; cur[3] = keep[3];
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
      // ghosts of the lines for the next sweep: the periodic image, or the homogeneous BC of the side
#pragma unroll
      for (int p = 0; p < 4; p++) {
	ghostL[p] = A.sgn[1] == 0. ? lastv[p] : A.sgn[1]*first[p];     // I = -1 (left side)
	ghostR[p] = A.sgn[0] == 0. ? first[p] : A.sgn[0]*lastv[p];     // I = n  (right side)
      }
      if (A.mirror) {
#pragma unroll
	for (int p = 0; p < 4; p++) {
	  A.un[A.L.idx (1, jl[p], kl[p])] = first[p];
	  A.un[A.L.idx (n, jl[p], kl[p])] = lastv[p];
	}
      }
      if (write_ghosts) {
	// x ghosts of the last BC application (the y and z ghost planes: patch_loop_ghosts_kernel)
#pragma unroll
	for (int p = 0; p < 4; p++) {
	  A.un[A.L.idx (n + 1, jl[p], kl[p])] = ghostR[p];
	  A.un[A.L.idx (0, jl[p], kl[p])] = ghostL[p];
	}
      }
    }
    if (A.stats && tid0 == 0)
      A.stats[2*(tile*SK_MAXF + sw) + 1] = __builtin_amdgcn_s_memrealtime ();
  }
}

// y and z ghost planes left by the last BC application of the loop, from the granules of sweep
// nsweeps - 2 (see skew_loop_ghosts_kernel); rows in this file's convention
__global__ void __launch_bounds__(256)
patch_loop_ghosts_kernel (SkewLoopArgs A)
{
  const int n = A.L.n, ntj = A.ntj;
  const long hstride = (long) SK_HROWS (n)*SK_T;
  const int sw = A.nsweeps - 2;
  const u64 * hbJ = A.hb + sw*A.hb_sweep, * hbK = hbJ + A.hb_words;
  const u64 * snJ = hbK + A.hb_words, * snK = snJ + A.hb_words;
  const int I = blockIdx.x*blockDim.x + threadIdx.x;     // 0 .. n-1
  const int c = blockIdx.y;                               // the other tangential index, 0 .. n-1
  const int plane = blockIdx.z;
  if (I >= n) return;
  const int T_ = c / SK_T, l = c % SK_T, lh = l >> 1;
  const long lastJ = (long) ((ntj - 1) + ntj*T_)*hstride, firstJ = (long) (0 + ntj*T_)*hstride;
  const long lastK = (long) (T_ + ntj*(ntj - 1))*hstride, firstK = (long) (T_ + ntj*0)*hstride;
  const long rowHb = (long) (I + lh)*SK_T + l, rowSn = (long) (I + lh + 7)*SK_T + l;
  u64 bits;
  long dst;
  double sg;
  switch (plane) {
  case 0: sg = A.sgn[2]; bits = sg == 0. ? hbJ[lastJ + rowHb] : snJ[firstJ + rowSn];     // ghost j = n + 1
    dst = A.L.idx (I + 1, n + 1, n - c); break;
  case 1: sg = A.sgn[3]; bits = sg == 0. ? snJ[firstJ + rowSn] : hbJ[lastJ + rowHb];     // ghost j = 0
    dst = A.L.idx (I + 1, 0, n - c); break;
  case 2: sg = A.sgn[4]; bits = sg == 0. ? hbK[lastK + rowHb] : snK[firstK + rowSn];     // ghost k = n + 1
    dst = A.L.idx (I + 1, n - c, n + 1); break;
  default: sg = A.sgn[5]; bits = sg == 0. ? snK[firstK + rowSn] : hbK[lastK + rowHb];    // ghost k = 0
    dst = A.L.idx (I + 1, n - c, 0);
  }
  const double v = __longlong_as_double ((long long) bits);
  A.un[dst] = sg == 0. ? v : sg*v;
}

// ---------------------------------------------------------------------------------------------
// natural <-> patch-skewed copies.  One block = one tile, one pair of K planes (b = 2 PB, 2 PB + 1)
// and a chunk of 16 rows: on the skewed side the block moves 16 rows x 256 B (8 lanes x 32 B), on
// the natural side 32 lines x (16 + 7) cells; the transposition goes through LDS.
// ---------------------------------------------------------------------------------------------
struct PatchPackArgs {
  Layout L;
  int ntj, RT;
  const double * src[3];
  double * dst[3];
  int narr;
  double * add;            // unpack: add the values to this natural array instead of storing them
  // pack: array 0 is not read but prolongated from the next coarser level (get_from_above,
  // src/poisson.c:1005-1042 = prolongate_kernel): the natural array of the level is then written
  // only where the BC application that follows reads it (the cells along the six sides)
  const double * coarse;
  double * nat;
  Layout Lc;
};

// get_from_above of fine cell (i, j, k): prolongate_kernel's expression (poisson_kernels.hip)
__device__ __forceinline__ double patch_prolong (const Layout & Lc, const double * __restrict__ vc,
						 int i, int j, int k)
{
  const int pi = (i + 1)/2, pj = (j + 1)/2, pk = (k + 1)/2;
  const long p = Lc.idx (pi, pj, pk);
  const double pv = vc[p];
  double h[3];
  const long off[3] = { 1, Lc.sy, Lc.sz };
#pragma unroll
  for (int cc = 0; cc < 3; cc++) {
    double g1 = vc[p + off[cc]] - 1.*pv;
    double g2 = vc[p - off[cc]] - 1.*pv;
    h[cc] = (g1 - g2)/2.;
  }
  const double rel[3] = { ((i & 1) ? -1. : 1.)/4., ((j & 1) ? -1. : 1.)/4., ((k & 1) ? -1. : 1.)/4. };
  double val = pv;
#pragma unroll
  for (int cc = 0; cc < 3; cc++)
    val += rel[cc]*h[cc];
  return val;
}

// get_from_above (src/poisson.c:1005-1042) of a whole level STRAIGHT INTO the patch-skewed layout of
// its relax loop: the initial guess of the loop is produced where the loop reads it, the natural
// array of the level only gets the cells along the six box sides (what the BC application that
// follows reads).  One wave = one tile and PR_ROWS rows of it; lane (A, B) owns the 2 x 2 lines of
// the loop kernels and writes their four cells of row rho = I + A + B: every store instruction of the
// wave is one contiguous KB of a row.  The 2 x 2 lines of a lane and two consecutive I are the eight
// children of ONE coarse cell -- (m + 1, n/2 - 8 P - A, n/2 - 8 Q - B), m = I >> 1 -- whose value and
// three half-differences the lane keeps for two rows: 7 loads (scattered over the 64 coarse lines of
// the wave, L1-resident for 32 rows) and 10 operations per 8 cells, 3 additions per cell.  Operand
// order of patch_prolong / prolongate_kernel: val = p; val += rel_x h_x; += rel_y h_y; += rel_z h_z,
// rel = -+ 1/4 (the products by -+ 1/4 are exact: formed once per coarse cell as 0.25 h, negated).
/* rows per wave x waves per workgroup (64 rows of a tile per workgroup): measured at 256^3 / 128^3, average of
   the four calls of a step: 16 x 4: 35-37 us, 8 x 8: 30.5, 4 x 16: 29.0, 32 x 2: 52 (a wave's rows are a chain of
   LDS reads -> 10 operations -> two 1 KB stores: more waves hide it) */
#ifndef PR_ROWS
#define PR_ROWS 4
#endif
#ifndef PR_WAVES
#define PR_WAVES 16
#endif

// The seven coarse values of a lane are read from an LDS copy of the coarse cells the workgroup needs (10 x 10
// coarse lines -- the 8 x 8 of the lanes + the lines beside them -- over the m range of its 64 rows + 1 on either
// side), filled with coalesced loads: read straight from global memory the 64 lanes of a load touch 64
// different coarse lines, and the address coalescer, not HBM, set the time of the kernel (60 us for 151 MB at
// 256^3).  PR_LDS = 0: the loads from global memory.
#ifndef PR_LDS
#define PR_LDS 1
#endif
#define PR_W (((PR_ROWS*PR_WAVES + PK_SKEW) >> 1) + 4)      /* coarse cells along x the rows of a workgroup touch */

__global__ void __launch_bounds__(64*PR_WAVES)
patch_prolong_kernel (PatchPackArgs A)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.y;
  const int R0 = blockIdx.x*PR_WAVES*PR_ROWS;
  const int r0 = R0 + wave*PR_ROWS;
  const int n = A.L.n;
  const int P = tile % A.ntj, Q = tile / A.ntj;
  const int PA = lane & 7, PB = lane >> 3, s = PA + PB;
  double * const dst = A.dst[0] + (long) tile*(A.RT + 2*SK_FP)*SK_NL + SK_FP*SK_NL + 2*lane;
  // the coarse line of the lane and its neighbours
  const int pj = n/2 - 8*P - PA, pk = n/2 - 8*Q - PB;
  const double * __restrict__ const vc = A.coarse;
#if !PR_LDS
  const long cline = A.Lc.idx (0, pj, pk);
  const long csy = A.Lc.sy, csz = A.Lc.sz;
#else
  (void) pj; (void) pk;
  __shared__ double cbuf[10][10][PR_W];
  // coarse cells along x: index m + 1 with m = I >> 1 for the I of the workgroup's rows, and one either side
  const int Ilo = R0 - PK_SKEW > 0 ? R0 - PK_SKEW : 0;
  const int cx0 = Ilo >> 1;                              /* = (m_lo + 1) - 1 */
  {
    const int jb = n/2 - 8*P - 8, kb = n/2 - 8*Q - 8;   /* line (jl, kl) of the buffer = coarse line (jb + jl, kb + kl) */
    const int xmax = A.Lc.n + 1;
    for (int e = threadIdx.x; e < 100*PR_W; e += 64*PR_WAVES) {
      const int line = e / PR_W, ix = e % PR_W;
      const int jl = line % 10, kl = line / 10;
      int cx = cx0 + ix;
      if (cx > xmax) cx = xmax;                          /* beyond the last ghost cell: never used */
      cbuf[kl][jl][ix] = vc[A.Lc.idx (cx, jb + jl, kb + kl)];
    }
  }
  __syncthreads ();
  const double * const cl = &cbuf[8 - PB][8 - PA][0] - cx0;     /* cl[m + 1] = vc[cline + m + 1] */
  constexpr int lsy = PR_W, lsz = 10*PR_W;
#endif
  if (r0 >= n + PK_SKEW) return;
  // natural lines of the lane (p = da + 2 db) and whether they lie along a box side
  const int j0 = n - (SK_T*P + 2*PA), k0 = n - (SK_T*Q + 2*PB);
  const bool side_j[2] = { j0 == n, j0 - 1 == 1 }, side_k[2] = { k0 == n, k0 - 1 == 1 };
  int cur_m = -1;
  double pv = 0., qx = 0., qy = 0., qz = 0.;
  const int r1 = r0 + PR_ROWS;
  for (int rho = r0; rho < r1; rho++) {
    const int I = rho - s;
    if ((unsigned) I >= (unsigned) n) continue;
    const int m = I >> 1;
    if (m != cur_m) {
      cur_m = m;
#if PR_LDS
      const double * const q = cl + (m + 1);
      pv = q[0];
      {
	const double g1 = q[1] - 1.*pv, g2 = q[-1] - 1.*pv;
	qx = 0.25*((g1 - g2)/2.);
      }
      {
	const double g1 = q[lsy] - 1.*pv, g2 = q[-lsy] - 1.*pv;
	qy = 0.25*((g1 - g2)/2.);
      }
      {
	const double g1 = q[lsz] - 1.*pv, g2 = q[-lsz] - 1.*pv;
	qz = 0.25*((g1 - g2)/2.);
      }
#else
      const long p = cline + (m + 1);
      pv = vc[p];
      {
	const double g1 = vc[p + 1] - 1.*pv, g2 = vc[p - 1] - 1.*pv;
	qx = 0.25*((g1 - g2)/2.);
      }
      {
	const double g1 = vc[p + csy] - 1.*pv, g2 = vc[p - csy] - 1.*pv;
	qy = 0.25*((g1 - g2)/2.);
      }
      {
	const double g1 = vc[p + csz] - 1.*pv, g2 = vc[p - csz] - 1.*pv;
	qz = 0.25*((g1 - g2)/2.);
      }
#endif
    }
    // i = I + 1 odd (I even): the first child along x, rel = -1/4; j even (a even): +1/4; k even: +1/4
    const double bx = pv + ((I & 1) ? qx : - qx);
    d2 o01, o23;
    {
      const double by0 = bx + qy, by1 = bx + (- qy);
      o01.x = by0 + qz;     o01.y = by1 + qz;
      o23.x = by0 + (- qz); o23.y = by1 + (- qz);
    }
    double * const row = dst + (long) rho*SK_NL;
    *(d2 *) row = o01;
    *(d2 *) (row + 128) = o23;
    if (__builtin_expect (I == 0 || I == n - 1 || side_j[0] || side_j[1] || side_k[0] || side_k[1], 0)) {
      const bool ends = I == 0 || I == n - 1;
      const double vv[4] = { o01.x, o01.y, o23.x, o23.y };
#pragma unroll
      for (int q = 0; q < 4; q++)
	if (ends || side_j[q & 1] || side_k[q >> 1])
	  A.nat[A.L.idx (I + 1, j0 - (q & 1), k0 - (q >> 1))] = vv[q];
    }
  }
}

#ifndef PP_ROWS
#define PP_ROWS 16
#endif
#define PP_SPAN (PP_ROWS + 7)     /* cells along I touched by PP_ROWS rows of the 8 lanes of a B */
/* rows per block of the restriction + copy and of the copy out of the layout: with 64 rows a block
   owns 64 consecutive cells of a natural line (512 contiguous bytes instead of 128: fewer partial
   cache lines at the ends of a block's share; 79 -> 64 us and 106 -> 93 us per call, A/B on one box) */
#ifndef PPR_ROWS
#define PPR_ROWS 64
#endif
#define PPR_SPAN (PPR_ROWS + 7)
#ifndef PPU_ROWS
#define PPU_ROWS 64
#endif
#define PPU_SPAN (PPU_ROWS + 7)

__global__ void __launch_bounds__(256)
patch_pack_kernel (PatchPackArgs A)
{
  // [array][line: a + 16 db][cell along I], padded
  __shared__ double buf[3][32][PP_SPAN + 1];
  const int tid = threadIdx.x;
  const int tile = blockIdx.z, PB = blockIdx.y, r0 = blockIdx.x*PP_ROWS;
  const int P = tile % A.ntj, Q = tile / A.ntj;
  const int n = A.L.n;
  const long tbase = (long) tile*(A.RT + 2*SK_FP)*SK_NL + SK_FP*SK_NL;
  // natural cells needed: I = rho - PA - PB, rho = r0 .. r0 + 15, PA = 0..7  =>  I0 = r0 - 7 - PB .. r0 + 15 - PB
  const int I0 = r0 - 7 - PB;
  for (int e = tid; e < 32*PP_SPAN; e += 256) {
    const int line = e / PP_SPAN, di = e % PP_SPAN;
    const int a = line & 15, db = line >> 4;
    const int I = I0 + di;
    if (I >= 0 && I < n) {
      const int j = n - (SK_T*P + a), k = n - (SK_T*Q + 2*PB + db);
      const long nidx = A.L.idx (I + 1, j, k);
      int q0 = 0;
      if (A.coarse) {
	const double v = patch_prolong (A.Lc, A.coarse, I + 1, j, k);
	buf[0][line][di] = v;
	if (I == 0 || I == n - 1 || j == 1 || j == n || k == 1 || k == n)
	  A.nat[nidx] = v;
	q0 = 1;
      }
      for (int q = q0; q < A.narr; q++)
	buf[q][line][di] = A.src[q][nidx];
    }
  }
  __syncthreads ();
  // skewed side: 16 rows x 32 doubles (8 lanes x 4 cells) = 512 elements, 2 per thread
  for (int e = tid; e < PP_ROWS*32; e += 256) {
    const int row = e / 32, col = e % 32;
    const int PA = col >> 2, p = col & 3;
    const int a = 2*PA + (p & 1), db = p >> 1;
    const int rho = r0 + row;
    const int I = rho - PA - PB;
    if (I >= 0 && I < n) {
      const long sidx = tbase + (long) rho*SK_NL + 128*(p >> 1) + 2*(PA + 8*PB) + (p & 1);
      for (int q = 0; q < A.narr; q++)
	A.dst[q][sidx] = buf[q][a + 16*db][I - I0];
    }
  }
}

// The residual of a level on its way down the V-cycle: get_from_below (src/poisson.c:1044-1068 =
// restrict_kernel) onto the next coarser level AND the copy into the skewed layout of the level's
// own relax loop, from one read of the natural array (the 2 x 2 lines of a lane and two consecutive
// cells along I are the eight children of a coarse cell).  The coarse values go to the natural array
// of the coarse level and, where that level runs on these kernels too, into its skewed rhs.
struct PatchRestrictArgs {
  Layout L, Lc;
  int ntj, RT;
  const double * src;      // natural residual of the fine level
  double * dst;            // skewed rhs of the fine level
  double * cnat;           // natural residual of the coarse level
  double * cskew;          // skewed rhs of the coarse level, or nullptr
  int cntj, cRT;
  unsigned dimension;
};

__global__ void __launch_bounds__(256)
patch_restrict_pack_kernel (PatchRestrictArgs A)
{
  __shared__ double buf[32][PPR_SPAN + 2];
  const int tid = threadIdx.x;
  const int tile = blockIdx.z, PB = blockIdx.y, r0 = blockIdx.x*PPR_ROWS;
  const int P = tile % A.ntj, Q = tile / A.ntj;
  const int n = A.L.n;
  const long tbase = (long) tile*(A.RT + 2*SK_FP)*SK_NL + SK_FP*SK_NL;
  const int I0 = r0 - 7 - PB;
  for (int e = tid; e < 32*(PPR_SPAN + 1); e += 256) {
    const int line = e / (PPR_SPAN + 1), di = e % (PPR_SPAN + 1);
    const int a = line & 15, db = line >> 4;
    const int I = I0 + di;
    if (I >= 0 && I < n) {
      const int j = n - (SK_T*P + a), k = n - (SK_T*Q + 2*PB + db);
      buf[line][di] = A.src[A.L.idx (I + 1, j, k)];
    }
  }
  __syncthreads ();
  for (int e = tid; e < PPR_ROWS*32; e += 256) {
    const int row = e / 32, col = e % 32;
    const int PA = col >> 2, p = col & 3;
    const int a = 2*PA + (p & 1), db = p >> 1;
    const int rho = r0 + row;
    const int I = rho - PA - PB;
    if (I >= 0 && I < n)
      A.dst[tbase + (long) rho*SK_NL + 128*(p >> 1) + 2*(PA + 8*PB) + (p & 1)] = buf[a + 16*db][I - I0];
  }
  // the coarse cells whose first child (I = 2 m) sits in one of this block's rows: 32 consecutive lanes take 32
  // consecutive m of one coarse line (PA), so that the stores to the coarse level are 256 contiguous bytes
  // (one lane per row and PA, half of them idle, scattered the stores over eight coarse lines per instruction)
  for (int e = tid; e < 8*(PPR_ROWS/2); e += 256) {
    const int PA = e / (PPR_ROWS/2), mi = e % (PPR_ROWS/2);
    const int par = (PA + PB + r0) & 1;            /* rows of the block whose I = r0 + row - PA - PB is even */
    const int row = 2*mi + par;
    const int I = r0 + row - PA - PB;
    if (I >= 0 && I < n) {
      const int di = I - I0;
      double val = 0.;
      // children in child-id order: bit 0 -> +x, bit 1 -> -y (da = 1), bit 2 -> -z (db = 1)
#pragma unroll
      for (int id = 0; id < 8; id++)
	val += buf[2*PA + ((id >> 1) & 1) + 16*(id >> 2)][di + (id & 1)];
      const double v = A.dimension == 2 ? val : val/2.;
      const int m = I >> 1;
      const int pj = (n - SK_T*P - 2*PA) >> 1, pk = (n - SK_T*Q - 2*PB) >> 1;
      A.cnat[A.Lc.idx (m + 1, pj, pk)] = v;
      if (A.cskew) {
	const int ca = 8*(P & 1) + PA, cb = 8*(Q & 1) + PB;
	const long ctile = (P >> 1) + (long) A.cntj*(Q >> 1);
	A.cskew[ctile*(A.cRT + 2*SK_FP)*SK_NL + SK_FP*SK_NL +
		(long) (m + (ca >> 1) + (cb >> 1))*SK_NL + 128*(cb & 1) + 2*((ca >> 1) + 8*(cb >> 1)) + (ca & 1)] = v;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
patch_unpack_kernel (PatchPackArgs A)
{
  __shared__ double buf[32][PPU_SPAN + 1];
  const int tid = threadIdx.x;
  const int tile = blockIdx.z, PB = blockIdx.y, r0 = blockIdx.x*PPU_ROWS;
  const int P = tile % A.ntj, Q = tile / A.ntj;
  const int n = A.L.n;
  const long tbase = (long) tile*(A.RT + 2*SK_FP)*SK_NL + SK_FP*SK_NL;
  // the natural cells this block owns: I = r0 - PB .. r0 + 15 - PB for lane PA = 0 ... every cell
  // (I, a, b) belongs to exactly one row rho = I + PA + PB, hence to exactly one block
  const int I0 = r0 - 7 - PB;
  for (int e = tid; e < PPU_ROWS*32; e += 256) {
    const int row = e / 32, col = e % 32;
    const int PA = col >> 2, p = col & 3;
    const int a = 2*PA + (p & 1), db = p >> 1;
    const int rho = r0 + row;
    const int I = rho - PA - PB;
    if (I >= 0 && I < n)
      buf[a + 16*db][I - I0] = A.src[0][tbase + (long) rho*SK_NL + 128*(p >> 1) + 2*(PA + 8*PB) + (p & 1)];
  }
  __syncthreads ();
  for (int e = tid; e < 32*PPU_SPAN; e += 256) {
    const int line = e / PPU_SPAN, di = e % PPU_SPAN;
    const int a = line & 15, db = line >> 4;
    const int I = I0 + di;
    // the cell is in this block's rows iff r0 <= I + (a >> 1) + PB < r0 + 16
    const int rho = I + (a >> 1) + PB;
    if (I >= 0 && I < n && rho >= r0 && rho < r0 + PPU_ROWS) {
      const int j = n - (SK_T*P + a), k = n - (SK_T*Q + 2*PB + db);
      const long c = A.L.idx (I + 1, j, k);
      if (A.add)
	A.add[c] += buf[line][di];     /* correct (src/poisson.c:998-1003) fused into the unpack */
      else
	A.dst[0][c] = buf[line][di];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool patch_regs ()
{
  return getenv ("GFSHIP_PATCH_REGS") != nullptr;
}

int patch_resident_per_cu ()
{
  int per_cu = 0;
  hipError_t e = patch_regs () ?
    hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, relax_patch_loop_kernel<true>, PK_NTHREADS, 0) :
    hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, relax_ring_loop_kernel<true, 0>, RK_NTHREADS, 0);
  int per_cu1 = 0;
  if (e == hipSuccess)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu1, relax_ring_loop_kernel<true, 1>, RK_NTHREADS, 0);
  if (per_cu1 < per_cu) per_cu = per_cu1;
  return e == hipSuccess ? per_cu : 0;
}

int patch_pack (gfship_domain * dom, int level, SkewPlan * S, const double * u, const double * rhs,
		const double * dia, const double * coarse)
{
  PatchPackArgs A;
  A.add = nullptr;
  A.coarse = coarse;
  A.nat = const_cast<double *> (u);
  A.Lc = dom->lay[level > 0 ? level - 1 : 0];
  A.L = dom->lay[level]; A.ntj = S->ntj; A.RT = S->RT;
  A.narr = 0;
  A.src[A.narr] = u;   A.dst[A.narr++] = S->us;
  if (rhs) { A.src[A.narr] = rhs; A.dst[A.narr++] = S->rs; }     /* nullptr: already there (patch_restrict_pack) */
  if (dia) { A.src[A.narr] = dia; A.dst[A.narr++] = S->ds; }
  const int rows = A.L.n + PK_SKEW + 1;
  if (coarse && !getenv ("GFSHIP_OLD_PROLONG_PACK")) {
    /* the prolongation straight into the layout (patch_prolong_kernel); arrays that still have to be
       copied (a dia that is not zero; the rhs when the restriction has not left it there) follow in
       the transposing copy */
    PatchPackArgs Pr = A;
    Pr.narr = 1;
    dim3 pgrid ((rows + PR_ROWS*PR_WAVES - 1)/(PR_ROWS*PR_WAVES), S->ntj*S->ntj);
    hipLaunchKernelGGL (patch_prolong_kernel, pgrid, dim3 (64*PR_WAVES), 0, dom->stream, Pr);
    GFSHIP_HIP (hipGetLastError ());
    if (A.narr == 1)
      return GFSHIP_OK;
    for (int q = 1; q < A.narr; q++) { A.src[q - 1] = A.src[q]; A.dst[q - 1] = A.dst[q]; }
    A.narr--;
    A.coarse = nullptr;
  }
  dim3 grid ((rows + PP_ROWS - 1)/PP_ROWS, 8, S->ntj*S->ntj);
  hipLaunchKernelGGL (patch_pack_kernel, grid, dim3 (256), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int patch_restrict_pack (gfship_domain * dom, int level, SkewPlan * S, const double * res,
			 double * res_coarse, SkewPlan * Sc, unsigned dimension)
{
  PatchRestrictArgs A;
  A.L = dom->lay[level]; A.Lc = dom->lay[level - 1];
  A.ntj = S->ntj; A.RT = S->RT;
  A.src = res; A.dst = S->rs;
  A.cnat = res_coarse;
  A.cskew = Sc ? Sc->rs : nullptr;
  A.cntj = Sc ? Sc->ntj : 0; A.cRT = Sc ? Sc->RT : 0;
  A.dimension = dimension;
  const int rows = A.L.n + PK_SKEW + 1;
  dim3 grid ((rows + PPR_ROWS - 1)/PPR_ROWS, 8, S->ntj*S->ntj);
  hipLaunchKernelGGL (patch_restrict_pack_kernel, grid, dim3 (256), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int patch_unpack (gfship_domain * dom, int level, SkewPlan * S, double * u, double * add_into)
{
  PatchPackArgs A;
  A.coarse = nullptr; A.nat = nullptr; A.Lc = dom->lay[level];
  A.L = dom->lay[level]; A.ntj = S->ntj; A.RT = S->RT;
  A.narr = 1;
  A.src[0] = S->us; A.dst[0] = u; A.add = add_into;
  const int rows = A.L.n + PK_SKEW + 1;
  dim3 grid ((rows + PPU_ROWS - 1)/PPU_ROWS, 8, S->ntj*S->ntj);
  hipLaunchKernelGGL (patch_unpack_kernel, grid, dim3 (256), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// launches the loop kernel (and the ghost planes for nrelax >= 2) with the arguments the caller has
// prepared (skew_loop_run, relax_skew_loop.hip: granules armed, ticket zeroed)
int patch_loop_launch (gfship_domain * dom, const SkewLoopArgs & A, int ntiles, bool has_dia,
		       unsigned nrelax, float * ms)
{
  if (ms) GFSHIP_HIP (hipEventRecord (dom->ev0, dom->stream));
  if (A.op == 1)            /* diffusion_relax: the ring kernel (rhoc travels as the dia stream) */
    hipLaunchKernelGGL ((relax_ring_loop_kernel<true, 1>), dim3 (ntiles), dim3 (RK_NTHREADS), 0, dom->stream, A);
  else if (patch_regs ()) {      /* the variant that streams through registers (GFSHIP_PATCH_REGS=1) */
    if (has_dia)
      hipLaunchKernelGGL (relax_patch_loop_kernel<true>, dim3 (ntiles), dim3 (PK_NTHREADS), 0, dom->stream, A);
    else
      hipLaunchKernelGGL (relax_patch_loop_kernel<false>, dim3 (ntiles), dim3 (PK_NTHREADS), 0, dom->stream, A);
  }
  else if (has_dia)
    hipLaunchKernelGGL ((relax_ring_loop_kernel<true, 0>), dim3 (ntiles), dim3 (RK_NTHREADS), 0, dom->stream, A);
  else
    hipLaunchKernelGGL ((relax_ring_loop_kernel<false, 0>), dim3 (ntiles), dim3 (RK_NTHREADS), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  if (ms) {
    GFSHIP_HIP (hipEventRecord (dom->ev1, dom->stream));
    GFSHIP_HIP (hipEventSynchronize (dom->ev1));
    GFSHIP_HIP (hipEventElapsedTime (ms, dom->ev0, dom->ev1));
  }
  if (nrelax >= 2) {
    const Layout & L = A.L;
    int block = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
    hipLaunchKernelGGL (patch_loop_ghosts_kernel, dim3 ((L.n + block - 1)/block, L.n, 4), dim3 (block),
			0, dom->stream, A);
    GFSHIP_HIP (hipGetLastError ());
  }
  return GFSHIP_OK;
}

} // namespace gfship
