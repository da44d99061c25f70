// relax_skew_loop.hip -- a whole relax_loop (src/poisson.c:1070-1089) of a periodic 3-D level as
// ONE launch: the sweeps of the loop are pipelined behind each other.
//
// One exact-order sweep has a dependency chain of 3n - 2 hyperplanes, during which most of the
// chip idles (only a diagonal band of tiles is active, relax_skew.hip).  The next sweep of the
// loop does not have to wait for the end of the previous one: cell (I,J,K) of sweep s + 1 needs
//   - the sweep-s values of its +1 neighbours (one hyperplane later in sweep s), and
//   - the homogeneous-BC ghosts, which on a periodic box are the sweep-s values of the cells on
//     the opposite side (gfs_domain_homogeneous_bc between the sweeps, src/poisson.c:1078-1086):
//     ghost I = -1 is cell I = n - 1 of the same line, so a line can start sweep s + 1 as soon as
//     it has finished sweep s.
// So every workgroup keeps its tile for the whole loop and starts sweep s + 1 right after its own
// sweep s: the loop takes one sweep plus (nrelax - 1) tile-sweeps (~ n steps each) instead of
// nrelax full sweeps (~ 3 n steps each).  Same arithmetic, same order per cell: bit-identical to
// the sweep-by-sweep schedule (tests/test_gpu_poisson.py compares both with the oracle).
//
// What sweep s + 1 reads from sweep s, and how it knows the data is there:
//   own line, one cell ahead ........ written by the same thread (program order)
//   ghost I = -1 / I = n ............. last / first value of the own line: registers
//   line (-1,b) / (a,-1) ............. hand-off granules of the neighbour tile, as in relax_skew.hip;
//                                      for the first tile row / column the periodic image comes
//                                      from the last tile row / column of sweep s (same granules,
//                                      written for the wrap as well)
//   line (16,b) / (a,16) ............. the first lines (a = 0 / b = 0) of the neighbour tile.  They
//                                      are overwritten in place by that tile's next sweep (the
//                                      periodic image tile is even AHEAD of the reader), so every
//                                      tile snapshots them per sweep into granule buffers
//                                      (data-is-flag, sentinel), read by the next sweep
// Non-periodic sides (Dirichlet, Neumann, symmetry; homogeneous between the sweeps) work the same
// way with the tile's OWN outgoing lines as sources: ghost line = +-(line next to the side), taken
// from the tile's snapshot (a = 0 / b = 0) or hand-off (a = 15 / b = 15) granules of the previous
// sweep, the sign applied when the value goes into LDS.
// Granule buffers exist once per sweep of the loop (no re-arming inside the launch).  Every
// wait is bounded and reported through *err.  All tiles must be resident at once (a tile of
// sweep s + 1 waits on tiles of sweep s): the host checks the occupancy and otherwise falls back
// to one launch per sweep.
#include "relax_skew.hpp"
#include <cstdlib>
#include <vector>

namespace gfship {


__device__ __forceinline__ unsigned skew_xcc_id ()
{
  unsigned v;
  asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s" (v));
  return v & 7;
}

__device__ __forceinline__ unsigned skew_claim_tile (const SkewLoopArgs & A)
{
  if (A.per_xcd == 0)
    return A.order[atomicAdd (A.ticket, 1u) + 1u]   /* the ticket is armed with the granules: all ones */;
  const unsigned x = skew_xcc_id ();
  for (unsigned q = 0; q < 8; q++) {
    const unsigned xx = (x + q) & 7;
    const unsigned k = atomicAdd (&A.xticket[xx], 1u);
    if (k < (unsigned) A.per_xcd)
      return A.xorder[xx*A.per_xcd + k];
  }
  return 0;     /* not reached: as many workgroups as tiles */
}

// Workgroup = 4 compute waves (256 lines) + 1 halo wave + 1 store wave.  The halo wave streams
// the four halo strips into LDS (and polls the granules that are not there yet); the store wave
// reads the new values of the four outgoing lines (a = 15, b = 15: hand-off; a = 0, b = 0:
// snapshot) from LDS one step after they were computed and writes the granules.  The compute
// waves, which all wait for the slowest of them at the barrier of every step, carry neither halo
// code nor agent-scope stores: those complete late, and since loads and stores share the in-order
// vmcnt counter they would hold back the prefetched loads behind them (tools/lab/step_lab.hip:
// two predicated sc1 stores cost a compute wave 0.09 us of a 0.15 us step).  (Moving the streaming
// loads and the row stores into further waves as well was tried and lost: the hops between tiles
// became slower and the step no faster.)
#define SK_NTHREADS (SK_NL + 128)
// OP == 2 (the six face weights f[d].v of every cell, gfs_poisson_coefficients with alpha): a seventh
// wave streams the rows of the six skewed weight arrays into a ring of SK_WR slots in LDS (LDS-DMA,
// global_load_lds: 1 KB per instruction, no registers), SK_WD steps ahead, and retires them with a
// counted s_waitcnt one step before the compute waves read them (the weights are not on the
// dependent chain of a step; as register streams they cost the compute waves 290 spilled VGPRs)
#define SK_WR 6              /* ring slots: divides the unroll factor SK_D of the step loop */
#define SK_WD 5              /* rows ahead (<= SK_WR - 1; 12*(SK_WD - 1) < 64: the vmcnt counter) */
#define SK_WBYTES (SK_WR*6*SK_NL*sizeof (double))
static_assert (SK_D % SK_WR == 0, "slot numbers of the weight ring must be compile-time constants");

template <bool HAS_DIA, int OP>
__global__ void __launch_bounds__(OP == 2 ? SK_NTHREADS + 64 : SK_NTHREADS)
relax_skew_loop_kernel (SkewLoopArgs A)
{
  constexpr int XS = SK_T + 1;
  __shared__ double X[2][XS*XS];
  __shared__ double Y[2][XS*XS];
  __shared__ unsigned s_tile;
  extern __shared__ double wring[];      // OP == 2: [SK_WR][6][SK_NL] face weights of the rows t .. t + SK_WD

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane (tid0 >> 6);                   // wave-uniform
  const bool compute = wave < SK_NL/64;
  const int tid = tid0 & (SK_NL - 1);      // line of a compute lane; lane index of the halo wave
  const int a = tid & (SK_T - 1), b = tid >> 4;
  const int n = A.L.n;
  const int ntj = A.ntj;
  const long tstride = (long) (A.RT + 2*SK_FP)*SK_NL;
  const long hstride = (long) SK_HROWS (n)*SK_T;

  if (tid0 == 0)
    s_tile = skew_claim_tile (A);
  __syncthreads ();
  const int tile = s_tile;
  const int P = tile % ntj, Q = tile / ntj;
  const int j = n - (SK_T*P + a), k = n - (SK_T*Q + b);
  const int s = a + b;
  // periodic neighbours of the tile
  const int tJm = (P > 0 ? P - 1 : ntj - 1) + ntj*Q, tJp = (P + 1 < ntj ? P + 1 : 0) + ntj*Q;
  const int tKm = P + ntj*(Q > 0 ? Q - 1 : ntj - 1), tKp = P + ntj*(Q + 1 < ntj ? Q + 1 : 0);

  double * const ut = A.us + tile*tstride + SK_FP*SK_NL;
  const bool loader = wave == SK_NL/64, storer = wave == SK_NL/64 + 1;
  const bool weigher = OP == 2 && wave == SK_NL/64 + 2;
#if SK_EXP & 4
  if (compute) __builtin_amdgcn_s_setprio (3);     /* experiment: compute waves first on their SIMD */
#endif
  const int g = (tid >> 4) & 3, m = tid & 15;
  const int jm = n - (SK_T*P + m), km = n - (SK_T*Q + m);
  bool failed = false;

  const int iOwnX = (a + 1) + XS*(b + 1), iT = a + XS*(b + 1), iF = (a + 1) + XS*b;
  const int iOwnY = a + XS*b, iBo = (a + 1) + XS*b, iBk = a + XS*(b + 1);
  const int T = (n + SK_PAD + SK_D - 1)/SK_D*SK_D;

  // ghost cells at the two ends of the line: natural ghosts for the first sweep, then the
  // periodic images kept in registers
  double ghostL = A.un[A.L.idx (0, j, k)];
  double ghostR = A.un[A.L.idx (n + 1, j, k)];

  for (int sw = 0; sw < A.nsweeps; sw++) {
    const bool more = sw + 1 < A.nsweeps;            // another sweep follows
    const bool write_ghosts = sw + 2 == A.nsweeps;   // its BC application is the last one
    u64 * const hbJ = A.hb + sw*A.hb_sweep, * const hbK = hbJ + A.hb_words;
    u64 * const snJ = hbK + A.hb_words, * const snK = snJ + A.hb_words;
    const u64 * const hbJp = hbJ - A.hb_sweep, * const hbKp = hbK - A.hb_sweep;   // previous sweep
    const u64 * const snJp = snJ - A.hb_sweep, * const snKp = snK - A.hb_sweep;

    // ---- own streams ----
    const double * qR = ut + SK_NL + tid;
    const double * qRhs = A.rs + tile*tstride + SK_FP*SK_NL + tid;
    const double * qDia = HAS_DIA ? A.ds + tile*tstride + SK_FP*SK_NL + tid : nullptr;
    // ---- halo streams of wave 0 ----
    const u64 * qH = A.dummy;
    int hs = 0;
    bool handoff = false;       // sentinel-guarded granule stream
    double hsgn = 1.;           // sign of a homogeneous BC whose ghost this stream carries
    int xy_halo = 0;
    if (loader) {
      switch (g) {
      case 0: // new values of line (-1, m)
	if (P > 0)       { qH = hbJ + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[2] == 0.) { qH = hbJp + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snJp + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[2]; }      /* own line a = 0, row t + 15 */
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - m, n + 1, km)); hs = 1; }
	xy_halo = 0 + XS*(m + 1);
	break;
      case 1: // new values of line (m, -1)
	if (Q > 0)       { qH = hbK + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[4] == 0.) { qH = hbKp + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snKp + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[4]; }      /* own line b = 0 */
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - m, jm, n + 1)); hs = 1; }
	xy_halo = (m + 1) + XS*0;
	break;
      case 2: // old values of line (16, m) = line (0, m) of tile (P+1, Q), row t - 15
	if (sw > 0 && (P + 1 < ntj || A.sgn[3] == 0.)) { qH = snJp + (long) tJp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbJp + (long) tile*hstride + m - (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[3]; }      /* own line a = 15, row t - 15 */
	else if (P + 1 < ntj) {
	  qH = (const u64 *) (A.us + tJp*tstride + SK_FP*SK_NL - (long) (SK_T - 1)*SK_NL + SK_T*m);
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (SK_T - 1 + m), 0, km)); hs = 1; }
	xy_halo = SK_T + XS*m;
	break;
      default: // old values of line (m, 16) = line (m, 0) of tile (P, Q+1), row t - 15
	if (sw > 0 && (Q + 1 < ntj || A.sgn[5] == 0.)) { qH = snKp + (long) tKp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbKp + (long) tile*hstride + m - (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[5]; }      /* own line b = 15 */
	else if (Q + 1 < ntj) {
	  qH = (const u64 *) (A.us + tKp*tstride + SK_FP*SK_NL - (long) (SK_T - 1)*SK_NL + m);
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (m + SK_T - 1), jm, 0)); hs = 1; }
	xy_halo = m + XS*SK_T;
      }
    }
    double * const halo_dst0 = (g < 2 ? &X[0][0] : &Y[0][0]) + xy_halo;
    const u64 * const qH0 = qH;
    // a granule stream is awaited only at the steps at which its consumer line is active:
    // strips 0/1 feed lines (0,m)/(m,0) (I = t - m), strips 2/3 feed lines (15,m)/(m,15)
    const int hlag = g < 2 ? m : m + SK_T - 1;

    // own row of the skewed copy (compute waves)
    double * wU = ut + tid;
    // store wave: lane (g, m) owns one outgoing line
    //   g = 0  line (15, m)  hand-off to tile (P+1, Q), also its periodic image when a sweep follows
    //   g = 1  line (m, 15)  hand-off to tile (P, Q+1)
    //   g = 2  line (0, m)   snapshot for the next sweep of tile (P-1, Q)
    //   g = 3  line (m, 0)   snapshot for the next sweep of tile (P, Q-1)
    u64 * pS = (u64 *) A.dummy;
    bool sOn = false;
    int sX = 0, sLag = 0;       // LDS index of the line's new value; skew a + b of the line
    double * pNat = nullptr;    // mirror mode: the line lies along a box side, natural address of I = -1 - sLag + t
    if (storer) {
      switch (g) {
      case 0: sOn = P + 1 < ntj || more; pS = hbJ + (long) tile*hstride + m - (long) (SK_T - 1)*SK_T;
	sX = SK_T + XS*(m + 1); sLag = SK_T - 1 + m;
	if (A.mirror && P == ntj - 1) pNat = A.un + A.L.idx (0, 1, km) - sLag;
	break;
      case 1: sOn = Q + 1 < ntj || more; pS = hbK + (long) tile*hstride + m - (long) (SK_T - 1)*SK_T;
	sX = (m + 1) + XS*SK_T; sLag = SK_T - 1 + m;
	if (A.mirror && Q == ntj - 1) pNat = A.un + A.L.idx (0, jm, 1) - sLag;
	break;
      case 2: sOn = more; pS = snJ + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T;
	sX = 1 + XS*(m + 1); sLag = m;
	if (A.mirror && P == 0) pNat = A.un + A.L.idx (0, n, km) - sLag;
	break;
      default: sOn = more; pS = snK + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T;
	sX = (m + 1) + XS*1; sLag = m;
	if (A.mirror && Q == 0) pNat = A.un + A.L.idx (0, jm, n) - sLag;
      }
    }

    double pR[SK_D], pRhs[SK_D], pDia[SK_D], pH[SK_DH];

#define SK_PREFETCH(q_)							\
    do {								\
      pR[q_]   = *qR;   qR += SK_NL;					\
      pRhs[q_] = *qRhs; qRhs += SK_NL;					\
      if (HAS_DIA) { pDia[q_] = *qDia; qDia += SK_NL; }			\
    } while (0)
#define SK_PREFETCH_HALO(q_)						\
    do {								\
      pH[q_] = __longlong_as_double ((long long) SK_HLOAD (qH)); qH += hs; \
    } while (0)
#define SK_HALO(t_, q_) SK_HALO_ (t_, q_, 0)
#define SK_HALO_STEP(t_, q_) SK_HALO_ (t_, q_, 1)
    // wave 0: halo value of step t_ (slot q_) into LDS buffer t_ & 1; refill_: then reload the slot
#define SK_HALO_(t_, q_, refill_)					\
    do {								\
      {									\
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
	  pH[q_] = __longlong_as_double ((long long) SK_HLOAD (qH)); qH += hs; \
	}								\
      }									\
    } while (0)

    if (A.stats && tid0 == 0)
      A.stats[2*(tile*SK_MAXF + sw)] = __builtin_amdgcn_s_memrealtime ();
    double h0 = 0.;
    if (loader) {
      h0 = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs;
#pragma unroll
      for (int q = 0; q < SK_DH; q++)
	SK_PREFETCH_HALO (q);
    }
    else if (compute) {
#pragma unroll
      for (int q = 0; q < SK_D; q++)
	SK_PREFETCH (q);
    }

    __syncthreads ();      // the LDS grids (and the weight ring) of the previous sweep are no longer read
    // weights wave: rows 0 .. SK_WD - 1 of the six arrays into their slots
    const int wlane = tid0 & 63;
    // running pointers of the six arrays (a row further per step: the addresses stay one addition away)
    const double * wp[6];
#define SK_WROW(slot_)							\
    do {								\
      _Pragma ("unroll")						\
      for (int d_ = 0; d_ < 6; d_++) {					\
	double * dst_ = wring + ((slot_)*6 + d_)*SK_NL;			\
	__builtin_amdgcn_global_load_lds ((const void *) wp[d_],	\
					  (__attribute__((address_space(3))) void *) dst_, 16, 0, 0); \
	__builtin_amdgcn_global_load_lds ((const void *) (wp[d_] + 128),	\
					  (__attribute__((address_space(3))) void *) (dst_ + 128), 16, 0, 0); \
	wp[d_] += SK_NL;						\
	asm volatile ("" : "+v" (wp[d_]));      /* six running pointers, not one per unrolled row */ \
      }									\
    } while (0)
    if (weigher) {
#pragma unroll
      for (int d = 0; d < 6; d++)
	wp[d] = A.ws[d] + tile*tstride + SK_FP*SK_NL + 2*wlane;
#pragma unroll
      for (int r = 0; r < SK_WD; r++)
	SK_WROW (r % SK_WR);
      asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int q = tid0; q < 2*XS*XS; q += SK_NTHREADS) {
      (&X[0][0])[q] = 0.;
      (&Y[0][0])[q] = 0.;
    }
    __syncthreads ();
    if (loader) {
      double keep = pH[0];
      pH[0] = h0;
      SK_HALO (0, 0);
      pH[0] = keep;
    }
    else if (compute)
      Y[0][iOwnY] = pR[0];
    __syncthreads ();

    double prev = ghostL, first = 0.;

    if (loader) {
      // ---- halo wave: one LDS write (+ one granule load) per step, same barriers ----
      for (int t0 = 0; t0 < T; t0 += SK_D) {
#pragma unroll
	for (int q = 0; q < SK_D; q++) {
	  const int t = t0 + q;
	  if (!(SK_KO & 2))
	    SK_HALO_STEP (t + 1, q % SK_DH);
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
    }
    else if (storer) {
      // ---- store wave: the value a line computed at step t - 1 sits in X[t & 1]; its granule
      //      belongs to row (t - 1) -+ 15 of the line's buffer ----
      for (int t0 = 0; t0 < T; t0 += SK_D) {
#pragma unroll
	for (int q = 0; q < SK_D; q++) {
	  const int t = t0 + q;
	  const int I = t - 1 - sLag;
	  if (pNat && I >= 0 && I < n)
	    pNat[t] = X[t & 1][sX];          /* cell i = I + 1 = t - sLag of the side line */
	  if (!(SK_KO & 1) && sOn && I >= 0 && I < n)
	    { if (SK_EXP & 2) *(pS - SK_T) = (u64) __double_as_longlong (X[t & 1][sX]);
	      else store_sc1 (pS - SK_T, (u64) __double_as_longlong (X[t & 1][sX])); }
	  pS += SK_T;
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
    }
    else if (weigher) {
      // ---- weights wave: row t + SK_WD into the slot the compute waves read at step t - 1 ----
      for (int t0 = 0; t0 < T; t0 += SK_D) {
#pragma unroll
	for (int q = 0; q < SK_D; q++) {
	  SK_WROW ((q + SK_WD) % SK_WR);
	  {
	    // rows up to t + 1 have landed: s_waitcnt vmcnt (12*(SK_WD - 1)) alone (gfx9 encoding
	    // vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
	    constexpr int N = 12*(SK_WD - 1);
	    static_assert (N < 64, "the vmcnt counter holds 63");
	    __builtin_amdgcn_s_waitcnt ((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
	  }
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
    }
    else {
      // ---- compute waves ----
      for (int t0 = 0; t0 < T; t0 += SK_D) {
#pragma unroll
	for (int q = 0; q < SK_D; q++) {
	  const int t = t0 + q;
	  const int I = t - s;
	  const bool act = I >= 0 && I < n;
	  const int B = t & 1;
	  const double Tn = X[B][iT], Fn = X[B][iF], Bo = Y[B][iBo], Bk = Y[B][iBk];
	  // relax, src/poisson.c:507-530, unit weights, d = 0..5 = right,left,top,bottom,front,back
	  const double Rv = (I + 1 < n) ? pR[q] : ghostR;
	  double aa = HAS_DIA ? pDia[q] : 0., bb = 0.;
	  aa += 1.; bb += 1.*Rv;
	  aa += 1.; bb += 1.*prev;
	  aa += 1.; bb += 1.*Tn;
	  aa += 1.; bb += 1.*Bo;
	  aa += 1.; bb += 1.*Fn;
	  aa += 1.; bb += 1.*Bk;
	  if (OP == 2) {
	    // relax with the face weights of the cell (src/poisson.c:507-530, face_weighted_gradient's
	    // same-level branch src/fluid.c:858-864: g.a = w, g.b = w*u_nb), d = 0..5
	    aa = pDia[q]; bb = 0.;
	    const double * const wl = wring + (q % SK_WR)*(6*SK_NL) + tid;      // slot of row t
	    { const double g = wl[0*SK_NL]; aa += g; bb += g*Rv; }
	    { const double g = wl[1*SK_NL]; aa += g; bb += g*prev; }
	    { const double g = wl[2*SK_NL]; aa += g; bb += g*Tn; }
	    { const double g = wl[3*SK_NL]; aa += g; bb += g*Bo; }
	    { const double g = wl[4*SK_NL]; aa += g; bb += g*Fn; }
	    { const double g = wl[5*SK_NL]; aa += g; bb += g*Bk; }
	  }
	  const double v = OP == 1 ? diffusion_cell (Rv, prev, Tn, Bo, Fn, Bk, pRhs[q], pDia[q], A.w, A.h2) :
	    HAS_DIA ? (aa != 0. ? (bb - pRhs[q])/aa : 0.) : divide_by_6 (bb - pRhs[q]);
	  prev = act ? v : prev;
	  first = I == 0 ? v : first;
	  X[B ^ 1][iOwnX] = v;
	  Y[B ^ 1][iOwnY] = pR[(q + 1) % SK_D];
	  if (!(SK_KO & 8))
	    SK_PREFETCH (q);
	  // own row of the skewed copy (read back by the same thread only)
	  if (!(SK_KO & 4))
	    *wU = v;
	  wU += SK_NL;
	  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
      }
    }
    // end of the sweep: the periodic line ghosts of the next one
    if (A.stats && tid0 == 0)
      A.stats[2*(tile*SK_MAXF + sw) + 1] = __builtin_amdgcn_s_memrealtime ();
    // ghosts of the line for the next sweep: the periodic image, or the homogeneous BC of the side
    {
      const double last = prev;                                  // value at I = n - 1
      ghostL = A.sgn[1] == 0. ? last : A.sgn[1]*first;         // I = -1 (left side)
      ghostR = A.sgn[0] == 0. ? first : A.sgn[0]*last;         // I = n  (right side)
    }
    if (A.mirror && compute) {
      A.un[A.L.idx (1, j, k)] = first;
      A.un[A.L.idx (n, j, k)] = prev;
    }
    if (write_ghosts && compute) {
      // the ghost layer the last BC application of the loop leaves behind (read by
      // get_from_above, poisson.c:1160-1167): the x ghosts of a line are only ever read by its
      // own thread (at the start of the launch), so they can be written here; the y and z ghost
      // planes are still read by tiles in their first sweep and are filled after the launch
      // from the granules of this sweep (skew_loop_ghosts_kernel)
      A.un[A.L.idx (n + 1, j, k)] = ghostR;
      A.un[A.L.idx (0, j, k)] = ghostL;
    }
#undef SK_WROW
#undef SK_PREFETCH
#undef SK_PREFETCH_HALO
#undef SK_HALO
#undef SK_HALO_STEP
#undef SK_HALO_
  }
}

// ---------------------------------------------------------------------------------------------
// Staggered waves: the exchange between neighbouring lines taken off LDS + barrier.
//
// The sweeps of a loop on a periodic box follow each other at the pace of the dependency chain
// across the box (the first tile row needs the hand-off of the last one: 15 hops per sweep), so a
// loop takes ~ 75 hops and a hop is ~ 21 steps + the hand-off latency: what counts is the latency
// of ONE step.  In the six-wave kernel above a step is LDS write -> barrier -> LDS read -> 10
// dependent fp64 operations (0.18 us without any memory instruction, tools/lab/step_lab.hip).
// Here wave w = 0..3 owns the lines b = 4w .. 4w+3, lane (a, p): a = lane & 15, p = lane >> 4,
// b = 4w + p (column a + 16 b = lane + 64 w: a wave's cells of a row are 512 contiguous bytes),
// and runs w steps behind wave 0: I = t - w - a - b.  Then
//   line (a, b-1), new .... ds_bpermute from lane - 16 of the same wave; for p = 0 the value wave
//                           w - 1 computed TWO steps earlier (one step of slack: through LDS, read
//                           a step ahead, off the chain) or the K- strip of the helper wave
//   line (a-1, b), new .... DPP row_shr:1 (rows of 16 lanes = one b); lane a = 0 keeps the DPP
//                           `old' operand = the J- strip value
//   line (a+1, b), old .... DPP row_shl:1 of the prefetched row, `old' = J+ strip value
//   line (a, b+1), old .... its own prefetched global load (same row, 16 columns on)
// so no barrier and no LDS round trip is on the chain of a step; the barrier that all waves still
// meet at once per step only orders the LDS buffers.  (One wave per tile with four lines per lane,
// tools/lab/wave_lab.hip, was tried first: a wave64 instruction takes 4 cycles on the 16-lane
// SIMD, 190 instructions per step = 0.37 us.)  The helper wave does what the halo and store
// waves do above, two steps ahead / one step behind.  Same layout, granules and ghost kernel.
// ---------------------------------------------------------------------------------------------
#ifndef WV_D
#define WV_D 16                /* prefetch distance of the own streams (steps) */
#endif
#ifndef WV_DH
#define WV_DH 8                /* prefetch distance of the helper wave's strips (steps) */
#endif
#define WV_TQ 16               /* the step count is a multiple of WV_D and WV_DH */
#define WV_NTHREADS 320
#ifndef WV_KO
#define WV_KO 0      /* timing experiments only: knock out parts of the step (wrong results) */
#endif
#define WV_BARRIER() asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WV_STEP_BARRIER() do { if (WV_KO & 8) asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory"); else WV_BARRIER (); } while (0)

// lanes without a source in the row keep oldv
template <int CTRL>
__device__ __forceinline__ double dpp_f64 (double oldv, double v)
{
  const long long bv = __double_as_longlong (v), ov = __double_as_longlong (oldv);
  const int lo = __builtin_amdgcn_update_dpp ((int) ov, (int) bv, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp ((int) (ov >> 32), (int) (bv >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double (((long long) hi << 32) | (unsigned) lo);
}

__device__ __forceinline__ double bperm_f64 (int addr, double v)
{
  const long long bv = __double_as_longlong (v);
  const int lo = __builtin_amdgcn_ds_bpermute (addr, (int) bv);
  const int hi = __builtin_amdgcn_ds_bpermute (addr, (int) (bv >> 32));
  return __longlong_as_double (((long long) hi << 32) | (unsigned) lo);
}

__global__ void __launch_bounds__(WV_NTHREADS)
relax_wave_loop_kernel (SkewLoopArgs A)
{
  // strips by g = 0..3, 16 values each: in  (-1,m) new, (m,-1) new, (16,m) old, (m,16) old
  //                                      out (15,m), (m,15) hand-off, (0,m), (m,0) snapshot
  __shared__ double H[4][64];     // slot t & 3: incoming strips of step t (strip 0 = XW[0])
  __shared__ double O[2][64];     // slot (t + 1) & 1: outgoing values of step t
  __shared__ double XW[4][4][16]; // [w][slot t & 3]: line (a, 4w - 1) for wave w at step t
  __shared__ unsigned s_tile;

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane (tid0 >> 6);
  const int lane = tid0 & 63;
  const int n = A.L.n;
  const int ntj = A.ntj;
  const long tstride = (long) (A.RT + 2*SK_FP)*SK_NL;
  const long hstride = (long) SK_HROWS (n)*SK_T;

  if (tid0 == 0)
    s_tile = A.order[atomicAdd (A.ticket, 1u) + 1u]   /* the ticket is armed with the granules: all ones */;
  __syncthreads ();
  const int tile = s_tile;
  const int P = tile % ntj, Q = tile / ntj;
  // local steps 0 .. n + 30 of wave 3 end at step n + 33; one more for the helper wave
  const int T = (n + SK_PAD + 4 + WV_TQ)/WV_TQ*WV_TQ;
  double * const ut = A.us + tile*tstride + SK_FP*SK_NL;

  // The compute waves and the helper wave run through separate code from here on (same number of
  // barriers): the counted s_waitcnt vmcnt of the software pipeline must not be merged with the
  // memory operations of the other role by the compiler.
  if (wave < 4) {
    // =========================== compute waves ===========================
    const int w = wave;
    const int a = lane & 15, p = lane >> 4, b = 4*w + p;
    const int j = n - (SK_T*P + a), k = n - (SK_T*Q + b);
    const int se = w + a + b;                      // I = t - se
    const int col = a + SK_T*b;                    // = lane + 64 w: the wave's cells of a row are contiguous
    const int up = ((lane - 16) & 63)*4;
    const bool lastB = b == SK_T - 1;              // line (a, 16) comes from the helper wave
    const double * const frontsrc = w == 0 ? &H[0][16] + a : &XW[w][0][0] + a;
    const int frontstride = w == 0 ? 64 : 16;      // doubles per slot

    double ghostL = A.un[A.L.idx (0, j, k)];
    double ghostR = A.un[A.L.idx (n + 1, j, k)];
    asm volatile ("" :: "v" (ghostL), "v" (ghostR));   // complete before the pipeline starts

    for (int sw = 0; sw < A.nsweeps; sw++) {
      const bool write_ghosts = sw + 2 == A.nsweeps;
      // wave w is at row t - w
      const double * qR = ut + (long) (1 - w)*SK_NL + col;
      const double * qRhs = A.rs + tile*tstride + (long) (SK_FP - w)*SK_NL + col;
      double * wU = ut - (long) w*SK_NL + col;
      double pR[WV_D], pB[WV_D], pRhs[WV_D];

      if (A.stats && tid0 == 0)
	A.stats[2*(tile*SK_MAXF + sw)] = __builtin_amdgcn_s_memrealtime ();
      // prologue with the memory instructions of WV_D steps (the stores go to unused rows in front
      // of the tile): same vmcnt distances as in the loop
#pragma unroll
      for (int q = 0; q < WV_D; q++) {
	pR[q] = qR[0]; pB[q] = qR[SK_T]; qR += SK_NL;
	pRhs[q] = *qRhs; qRhs += SK_NL;
	*(wU - (long) (WV_D - q)*SK_NL) = 0.;
	asm volatile ("" ::: "memory");          /* keep the order */
      }
      WV_BARRIER ();       // the LDS buffers of the previous sweep are no longer read
      WV_BARRIER ();       // strips of steps 0 and 1 are in H

      double prev = ghostL, first = 0., N = 0.;
      double frontH = frontsrc[0], hJm = H[0][b], hJp = H[0][32 + b], hKp = H[0][48 + a];

      for (int t0 = 0; t0 < T; t0 += WV_D) {
#pragma unroll
	for (int q = 0; q < WV_D; q++) {
	  const int t = t0 + q;
	  // values from outside the wave for the next step (written before the last barrier)
	  const double nfrontH = frontsrc[((t + 1) & 3)*frontstride];
	  const double * Hn = H[(t + 1) & 3];
	  const double nJm = Hn[b], nJp = Hn[32 + b], nKp = Hn[48 + a];
	  const double Rn = pR[q];               // own cell of row t - w + 1
	  const int I = t - se;
	  const bool act = I >= 0 && I < n;
	  double front = (WV_KO & 16) ? N : bperm_f64 (up, N);
	  front = p == 0 ? frontH : front;
	  const double top = (WV_KO & 32) ? hJm : dpp_f64<0x111> (hJm, N);        /* row_shr:1, lane a = 0 keeps the strip */
	  const double bot = (WV_KO & 32) ? hJp : dpp_f64<0x101> (hJp, Rn);       /* row_shl:1, lane a = 15 keeps the strip */
	  const double back = lastB ? hKp : pB[q];
	  // relax, src/poisson.c:507-530, unit weights, d = 0..5 = right,left,top,bottom,front,back
	  const double Rv = (I + 1 < n) ? Rn : ghostR;
	  double bb = 0.;
	  bb += 1.*Rv;
	  bb += 1.*prev;
	  bb += 1.*top;
	  bb += 1.*bot;
	  bb += 1.*front;
	  bb += 1.*back;
	  const double v = divide_by_6 (bb - pRhs[q]);
	  prev = act ? v : prev;
	  first = I == 0 ? v : first;
	  N = v;
	  // for the next wave (two steps later) and the helper wave (next step)
	  if (!(WV_KO & 2)) {
	    double * const Ow = O[(t & 1) ^ 1];
	    if (p == 3) {
	      if (w == 3) Ow[16 + a] = v;
	      else XW[w + 1][(t + 2) & 3][a] = v;
	    }
	    if (a == SK_T - 1) Ow[b] = v;
	    if (a == 0) Ow[32 + b] = v;
	    if (b == 0) Ow[48 + a] = v;
	  }
	  if (!(WV_KO & 4)) {
	    pR[q] = qR[0]; pB[q] = qR[SK_T]; qR += SK_NL;
	    pRhs[q] = *qRhs; qRhs += SK_NL;
	    *wU = v; wU += SK_NL;
	  }
	  frontH = nfrontH; hJm = nJm; hJp = nJp; hKp = nKp;
	  WV_STEP_BARRIER ();
	}
      }
      if (A.stats && tid0 == 0)
	A.stats[2*(tile*SK_MAXF + sw) + 1] = __builtin_amdgcn_s_memrealtime ();
      // ghosts of the line for the next sweep: the periodic image, or the homogeneous BC of the side
      {
	const double last = prev;
	ghostL = A.sgn[1] == 0. ? last : A.sgn[1]*first;
	ghostR = A.sgn[0] == 0. ? first : A.sgn[0]*last;
      }
      if (write_ghosts) {
	A.un[A.L.idx (n + 1, j, k)] = ghostR;
	A.un[A.L.idx (0, j, k)] = ghostL;
      }
    }
  }
  else {
    // =========================== helper wave ===========================
    const int g = lane >> 4, m = lane & 15;        // strip g, line m of the strip
    const int tJm = (P > 0 ? P - 1 : ntj - 1) + ntj*Q, tJp = (P + 1 < ntj ? P + 1 : 0) + ntj*Q;
    const int tKm = P + ntj*(Q > 0 ? Q - 1 : ntj - 1), tKp = P + ntj*(Q + 1 < ntj ? Q + 1 : 0);
    const int jm = n - (SK_T*P + m), km = n - (SK_T*Q + m);
    // the wave that consumes / produces the line of this lane runs woff steps behind wave 0
    const int wIn = g == 1 ? 0 : g == 3 ? 3 : m >> 2;
    const int wOut = g == 1 ? 3 : g == 3 ? 0 : m >> 2;
    const int hlag = (g < 2 ? m : m + SK_T - 1) + wIn;
    bool failed = false;

    for (int sw = 0; sw < A.nsweeps; sw++) {
      const bool more = sw + 1 < A.nsweeps;
      u64 * const hbJ = A.hb + sw*A.hb_sweep, * const hbK = hbJ + A.hb_words;
      u64 * const snJ = hbK + A.hb_words, * const snK = snJ + A.hb_words;
      const u64 * const hbJp = hbJ - A.hb_sweep, * const hbKp = hbK - A.hb_sweep;
      const u64 * const snJp = snJ - A.hb_sweep, * const snKp = snK - A.hb_sweep;

      // ---- incoming strips: same sources as in relax_skew_loop_kernel ----
      const u64 * qH = A.dummy;
      int hs = 0;
      bool handoff = false;
      double hsgn = 1.;
      switch (g) {
      case 0:
	if (P > 0)       { qH = hbJ + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[2] == 0.) { qH = hbJp + (long) tJm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snJp + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[2]; }
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - m, n + 1, km)); hs = 1; }
	break;
      case 1:
	if (Q > 0)       { qH = hbK + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0 && A.sgn[4] == 0.) { qH = hbKp + (long) tKm*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = snKp + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[4]; }
	else             { qH = (const u64 *) (A.un + A.L.idx (1 - m, jm, n + 1)); hs = 1; }
	break;
      case 2:
	if (sw > 0 && (P + 1 < ntj || A.sgn[3] == 0.)) { qH = snJp + (long) tJp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbJp + (long) tile*hstride + m - (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[3]; }
	else if (P + 1 < ntj) {
	  qH = (const u64 *) (A.us + tJp*tstride + SK_FP*SK_NL - (long) (SK_T - 1)*SK_NL + SK_T*m);
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (SK_T - 1 + m), 0, km)); hs = 1; }
	break;
      default:
	if (sw > 0 && (Q + 1 < ntj || A.sgn[5] == 0.)) { qH = snKp + (long) tKp*hstride + m; hs = SK_T; handoff = true; }
	else if (sw > 0) { qH = hbKp + (long) tile*hstride + m - (long) (SK_T - 1)*SK_T; hs = SK_T;
			   handoff = true; hsgn = A.sgn[5]; }
	else if (Q + 1 < ntj) {
	  qH = (const u64 *) (A.us + tKp*tstride + SK_FP*SK_NL - (long) (SK_T - 1)*SK_NL + m);
	  hs = SK_NL;
	}
	else { qH = (const u64 *) (A.un + A.L.idx (1 - (m + SK_T - 1), jm, 0)); hs = 1; }
      }
      const u64 * const qH0 = qH;
      // row of the stream for step t_ of wave 0: the consumer wave is at its step t_ - wIn (the rows
      // in front of the first one are never used: clamped)
#define WV_ROW(t_) (qH0 + (long) ((t_) - wIn > 0 ? (t_) - wIn : 0)*hs)

      // ---- outgoing lines: granule of cell I of the line = row I + m (hand-off) / I + m + 15 (snapshot) ----
      u64 * pS;
      bool sOn;
      switch (g) {
      case 0: sOn = P + 1 < ntj || more; pS = hbJ + (long) tile*hstride + m; break;
      case 1: sOn = Q + 1 < ntj || more; pS = hbK + (long) tile*hstride + m; break;
      case 2: sOn = more; pS = snJ + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T; break;
      default: sOn = more; pS = snK + (long) tile*hstride + m + (long) (SK_T - 1)*SK_T;
      }
      const int sLag = (g < 2 ? SK_T - 1 + m : m) + wOut;   // I = t - 1 - sLag for the value in O[t & 1]
      pS += (long) m*SK_T;                                   // row I + m

      // strip value hv_ of step t_ into the ring (waits for a granule that is not there yet)
#define WV_HALO_PUT(t_, hv_)						\
      do {								\
	double hv = (hv_);						\
	bool wt = handoff && !failed && (unsigned) ((t_) - hlag) < (unsigned) n && \
	  (u64) __double_as_longlong (hv) == SK_SENTINEL;		\
	if (__builtin_expect (__any (wt), 0)) {				\
	  unsigned spins = 0;						\
	  _Pragma ("nounroll")						\
	  while (__any (wt)) {						\
	    __builtin_amdgcn_s_sleep (SK_POLL_SLEEP);			\
	    if (wt) {							\
	      hv = __longlong_as_double ((long long) load_sc1 (WV_ROW (t_))); \
	      wt = (u64) __double_as_longlong (hv) == SK_SENTINEL;	\
	    }								\
	    if (++spins > (1u << 18)) { *A.err = 1; failed = true; break; } \
	  }								\
	}								\
	H[(t_) & 3][lane] = hv*hsgn;					\
      } while (0)

      double pH[WV_DH];
      const double h0 = __longlong_as_double ((long long) load_sc1 (WV_ROW (0)));
      const double h1 = __longlong_as_double ((long long) load_sc1 (WV_ROW (1)));
#pragma unroll
      for (int q = 0; q < WV_DH; q++)
	pH[q] = __longlong_as_double ((long long) load_sc1 (WV_ROW (2 + q)));
      WV_BARRIER ();       // the LDS buffers of the previous sweep are no longer read
      WV_HALO_PUT (0, h0);
      WV_HALO_PUT (1, h1);
      WV_BARRIER ();

      for (int t0 = 0; t0 < T; t0 += WV_DH) {
#pragma unroll
	for (int q = 0; q < WV_DH; q++) {
	  const int t = t0 + q;
	  if (!(WV_KO & 1)) {
	    // the value a line computed at step t - 1 sits in O[t & 1]
	    const int I = t - 1 - sLag;
	    if (sOn && I >= 0 && I < n)
	      store_sc1 (pS + (long) I*SK_T, (u64) __double_as_longlong (O[t & 1][lane]));
	    // strips of step t + 2
	    WV_HALO_PUT (t + 2, pH[q]);
	    pH[q] = __longlong_as_double ((long long) load_sc1 (WV_ROW (t + 2 + WV_DH)));
	  }
	  WV_STEP_BARRIER ();
	}
      }
#undef WV_HALO_PUT
#undef WV_ROW
    }
  }
}

// y and z ghost planes left by the last BC application of the loop = periodic images (or, at a
// non-periodic side, +- the adjacent line) of the side cells after sweep nsweeps - 2, taken from
// that sweep's granules:
//   hand-off J of tile (ntj-1,Q): line a = 15 (j = 1)  -> ghost j = n + 1     row I + b
//   snapshot J of tile (0,Q):     line a = 0  (j = n)  -> ghost j = 0         row I + b + 15
//   hand-off K of tile (P,ntj-1): line b = 15 (k = 1)  -> ghost k = n + 1     row I + a
//   snapshot K of tile (P,0):     line b = 0  (k = n)  -> ghost k = 0         row I + a + 15
__global__ void __launch_bounds__(256)
skew_loop_ghosts_kernel (SkewLoopArgs A)
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
  const int T_ = c / SK_T, l = c % SK_T;                  // tile and line of the tangential index
  // periodic: the line next to the opposite side; otherwise sgn * the line next to the same side
  const long lastJ = (long) ((ntj - 1) + ntj*T_)*hstride, firstJ = (long) (0 + ntj*T_)*hstride;
  const long lastK = (long) (T_ + ntj*(ntj - 1))*hstride, firstK = (long) (T_ + ntj*0)*hstride;
  const long rowHb = (long) (I + l)*SK_T + l, rowSn = (long) (I + l + SK_T - 1)*SK_T + l;
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
// host side
// ---------------------------------------------------------------------------------------------

// workgroups of the loop kernel of a level that are resident at once on the device
static int skew_loop_resident (gfship_domain * dom, int level)
{
  if (dom->skew_resident < 0) {
    int per_cu = 0, per_cu_w = 0, per_cu_2 = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice (&dev) != hipSuccess || hipGetDeviceProperties (&prop, dev) != hipSuccess ||
	hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu, relax_skew_loop_kernel<true, 1>,
						      SK_NTHREADS, 0) != hipSuccess ||
	hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu_w, relax_wave_loop_kernel,
						      WV_NTHREADS, 0) != hipSuccess ||
	hipFuncSetAttribute ((const void *) relax_skew_loop_kernel<true, 2>,
			     hipFuncAttributeMaxDynamicSharedMemorySize, SK_WBYTES) != hipSuccess ||
	hipOccupancyMaxActiveBlocksPerMultiprocessor (&per_cu_2, relax_skew_loop_kernel<true, 2>,
						      SK_NTHREADS + 64, SK_WBYTES) != hipSuccess) {
      dom->skew_resident = 0;
      dom->patch_resident = 0;
    }
    else {
      if (per_cu_2 < per_cu) per_cu = per_cu_2;
      dom->skew_resident = (per_cu < per_cu_w ? per_cu : per_cu_w)*prop.multiProcessorCount;
      dom->patch_resident = patch_resident_per_cu ()*prop.multiProcessorCount;
    }
  }
  return patch_level (dom, level) ? dom->patch_resident : dom->skew_resident;
}

bool skew_loop_supported (gfship_domain * dom, int level, unsigned nrelax, bool bc)
{
  if (!bc || nrelax < 2 || nrelax > SK_MAXF) return false;
  if (!skew_supported (dom, level)) return false;
  for (int d = 0; d < 6; d++)
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL) return false;   /* the exchange comes between the sweeps */
  // every tile waits on tiles of the previous sweep: all of them must be resident
  int ntj = dom->lay[level].n/SK_T, ntiles = ntj*ntj;
  return ntiles <= skew_loop_resident (dom, level);
}

// geometry of the granule sets of a level: words of the header, of a set, and of what a loop of
// nrelax sweeps needs armed
struct SetGeometry { long hb_words, hb_sweep; size_t hdr, set_words, need; };

static SetGeometry skew_set_geometry (const gfship_domain * dom, int level, const SkewPlan * S, unsigned nrelax)
{
  const Layout & L = dom->lay[level];
  const int ntiles = S->ntj*S->ntj;
  SetGeometry G;
  const long hstride = (long) SK_HROWS (L.n)*SK_T;
  G.hb_words = (long) ntiles*hstride;
  G.hb_sweep = 4*G.hb_words;
  /* header: [0] ticket, [1..4] the 8 tickets of the XCD blocks, [8..] the XCD of each tile (32 bits each) */
  G.hdr = 8 + ((size_t) ntiles + 1)/2 + 8;
  G.set_words = (size_t) SK_MAXF*G.hb_sweep + G.hdr + (G.hdr & 1);
  /* a single sweep only uses the two hand-off arrays of its granule set; the snapshots of the last
     sweep are neither written nor read: its two hand-off arrays end the armed range */
  G.need = G.hdr + (G.hdr & 1) + (size_t) (nrelax - 1)*G.hb_sweep + (size_t) 2*G.hb_words;
  return G;
}

// The granules of the set the NEXT loop of `level' will use, armed from this point of the main
// stream on the side stream (poisson_cycle calls this on the way down the V-cycle: the fill -- 148 MB
// at 256^3 -- then runs beside the relax loops of the coarser levels, which are bound by the latency
// of their hand-off chains and leave the memory system idle; in line it costs 37 us per loop at
// 256^3).  The loop waits for the event.  Arming beside the loop of the level ITSELF was tried in
// round 2 and lost what it gained (its stores delay the hand-offs); this is a different place.
int skew_arm_ahead (gfship_domain * dom, int level, unsigned nrelax)
{
  SkewPlan * S = &dom->skew[level];
  if (dom->no_arm_ahead || !dom->side_stream || !S->hbf || S->arm_wait || nrelax < 2 || nrelax > SK_MAXF)
    return GFSHIP_OK;           /* nothing allocated yet (first loop of the level): armed in line */
  if (dom->no_fused_loop || !skew_loop_supported (dom, level, nrelax, true))
    return GFSHIP_OK;
  if (patch_level (dom, level) && !dom->no_kernel_arming)
    return GFSHIP_OK;           /* the loop kernels arm the other set themselves (opt-in) */
  const SetGeometry G = skew_set_geometry (dom, level, S, nrelax);
  const int set = S->cur_set;
  if (S->armed[set] >= G.need)
    return GFSHIP_OK;
  if (!S->arm_ev)
    GFSHIP_HIP (hipEventCreateWithFlags (&S->arm_ev, hipEventDisableTiming));
  u64 * const base = (u64 *) S->hbf + (size_t) set*G.set_words;
  GFSHIP_HIP (hipEventRecord (dom->side_fork, dom->stream));
  GFSHIP_HIP (hipStreamWaitEvent (dom->side_stream, dom->side_fork, 0));
  GFSHIP_HIP (hipMemsetAsync (base, 0xFF, G.need*sizeof (u64), dom->side_stream));
  GFSHIP_HIP (hipEventRecord (S->arm_ev, dom->side_stream));
  S->armed[set] = G.need;
  S->arm_wait = true;
  return GFSHIP_OK;
}

// nrelax >= 2: the fused loop of a periodic level.  nrelax == 1: one sweep of any level the
// pipelined sweep runs on (any sides; the BC kernel is applied around it by the caller, so the
// cells next to the box sides are mirrored into the natural array).
int skew_loop_run (gfship_domain * dom, int level, SkewPlan * S, double * u_nat, bool has_dia,
		   unsigned nrelax, float * ms, const Field * ubc, const RelaxOp * op)
{
  const Layout & L = dom->lay[level];
  int ntiles = S->ntj*S->ntj;
  GFSHIP_CHECK (!op || op->kind == 0 || ((op->kind == 1 || op->kind == 2) && has_dia), GFSHIP_EUNSUPPORTED,
		"the pipelined sweep knows the Poisson (unit or per-face weights) and the diffusion cell updates");
  GFSHIP_CHECK (!op || op->kind != 2 || (S->ws[0] && !patch_level (dom, level)), GFSHIP_EINVAL,
		"weighted sweep without the skewed copies of the face weights");
  /* 8 words in front of the granules: the ticket counter, armed with them (it then counts from all
     ones: the claims add one).  Two sets of granules used in turn.  With GFSHIP_KERNEL_ARMING=1 the
     2 x 2 loop kernels arm the other set for the next loop of this level themselves, while their
     tiles wait for their first hand-off -- the fill before the launch (148 MB, 37 us at 256^3)
     disappears, but measured on the same box the loop gets slower by about as much (the arming
     stores delay the hand-offs of the pipeline's fill phase): off by default. */
  const SetGeometry G = skew_set_geometry (dom, level, S, nrelax);
  const long hb_words = G.hb_words, hb_sweep = G.hb_sweep;
  const size_t hdr = G.hdr, set_words = G.set_words, need = G.need;
  if (!S->hbf) {
    GFSHIP_HIP (hipMalloc ((void **) &S->hbf, 2*set_words*sizeof (u64)));
    S->armed[0] = S->armed[1] = 0;
    S->cur_set = 0;
  }
  const int set = S->cur_set;
  u64 * const base = (u64 *) S->hbf + (size_t) set*set_words;
  u64 * const other = (u64 *) S->hbf + (size_t) (set ^ 1)*set_words;
  if (S->arm_wait) {
    /* armed ahead on the side stream (skew_arm_ahead): the loop comes after that fill */
    GFSHIP_HIP (hipStreamWaitEvent (dom->stream, S->arm_ev, 0));
    S->arm_wait = false;
  }
  if (S->armed[set] < need)
    GFSHIP_HIP (hipMemsetAsync (base, 0xFF, need*sizeof (u64), dom->stream));
  const bool arms = patch_level (dom, level) && !dom->no_kernel_arming;
  if (arms && !S->arm_cum) {
    /* tile (P, Q) starts about P + Q hops after the first: its share of the arming grows with that
       slack, the tiles of the first diagonals get none */
    std::vector<unsigned> cum (ntiles + 1, 0);
    for (int t = 0; t < ntiles; t++) {
      const int w = t % S->ntj + t / S->ntj - 3;
      cum[t + 1] = cum[t] + (w > 0 ? w : 0);
    }
    if (cum[ntiles] == 0)
      for (int t = 0; t < ntiles; t++) cum[t + 1] = t + 1;
    GFSHIP_HIP (hipMalloc ((void **) &S->arm_cum, (ntiles + 1)*sizeof (unsigned)));
    GFSHIP_HIP (hipMemcpy (S->arm_cum, cum.data (), (ntiles + 1)*sizeof (unsigned), hipMemcpyHostToDevice));
  }
  S->armed[set] = 0;
  S->armed[set ^ 1] = arms ? need : 0;
  S->cur_set = set ^ 1;
  SkewLoopArgs A;
  A.L = L; A.ntj = S->ntj; A.RT = S->RT; A.nsweeps = (int) nrelax;
  A.mirror = nrelax == 1;
  for (int d = 0; d < 6; d++) {
    /* homogeneous BC of side d: symmetry src/boundary.c:45-51, dirichlet :253-268, neumann :336-347 */
    double sg = 0.;
    if (dom->side[d] != GFSHIP_SIDE_PERIODIC && ubc)
      switch (ubc->bc[d]) {
      case GFSHIP_BC_DIRICHLET: sg = -1.; break;
      case GFSHIP_BC_NEUMANN: sg = 1.; break;
      default: sg = ubc->component == d/2 ? -1. : 1.;
      }
    A.sgn[d] = sg;
  }
  A.us = S->us; A.rs = S->rs; A.ds = has_dia ? S->ds : nullptr;
  A.un = u_nat;
  A.hb = base + hdr + (hdr & 1); A.hb_sweep = hb_sweep; A.hb_words = hb_words;
  A.tile_xcd = (unsigned *) (base + 8);
  A.arm = other; A.arm_pairs = arms ? need/2 : 0; A.arm_cum = S->arm_cum;
  A.order = S->order;
  A.xorder = S->xorder;
  A.xticket = (unsigned *) S->ctl + 6;
  A.per_xcd = 0;
  A.near_mode = 0;
  A.op = op ? op->kind : 0;
  A.w = op ? op->w : 1.;
  A.h2 = op ? op->h2 : 1.;
  for (int d = 0; d < 6; d++) A.ws[d] = S->ws[d];
  { const char * e = getenv ("GFSHIP_FAULT_DROP_HANDOFF"); A.fault_tile = e ? atoi (e) : -1; }
  if (patch_level (dom, level)) {
    /* the 2 x 2 kernels, GFSHIP_XCD_SCOPE=1: XCD blocks + narrower-scope stores towards consumers on
       the same XCD (tickets in the armed header).  Measured at 256^3 on one box: blocks alone 0.51-0.52
       ms per loop against 0.505 without; + workgroup-scope stores (GFSHIP_XCD_NEAR_MODE=2) 0.485-0.52;
       + plain stores (=1) 1.04 (they linger in the L1's write path): the hand-off latency is not
       the store's way to memory.  Off by default. */
    if (!dom->no_xcd_scope && S->xorder && ntiles >= 8 && ntiles % 8 == 0 &&
	skew_loop_resident (dom, level) >= ntiles) {
      A.per_xcd = ntiles/8;
      A.xticket = (unsigned *) (base + 1);
      { const char * e = getenv ("GFSHIP_XCD_NEAR_MODE"); A.near_mode = e ? atoi (e) : 2; }
    }
  }
  else if (dom->xcd_place && S->xorder && ntiles >= 8 && skew_loop_resident (dom, level) >= ntiles) {
    A.per_xcd = ntiles/8;
    GFSHIP_HIP (hipMemsetAsync ((unsigned *) S->ctl + 6, 0, 8*sizeof (unsigned), dom->stream));
  }
  A.ticket = (unsigned *) base;
  A.err = (unsigned *) (dom->h_pinned + 32) + level;     /* skew_err_word, relax_skew.hip */
  A.dummy = (const u64 *) S->ctl + 2;
  A.stats = nullptr;
  if (getenv ("GFSHIP_SKEW_STATS")) {
    if (!S->stats_loop)
      GFSHIP_HIP (hipMalloc ((void **) &S->stats_loop, (size_t) ntiles*SK_MAXF*2*sizeof (u64)));
    A.stats = (u64 *) S->stats_loop;
  }
  if (patch_level (dom, level)) {
    int r = patch_loop_launch (dom, A, ntiles, has_dia, nrelax, ms);
    if (r) return r;
    if (nrelax >= 2 && A.stats && ms)
      goto dump_stats;
    return GFSHIP_OK;
  }
  if (ms) GFSHIP_HIP (hipEventRecord (dom->ev0, dom->stream));
  if (A.op == 1)                                     /* diffusion_relax */
    hipLaunchKernelGGL ((relax_skew_loop_kernel<true, 1>), dim3 (ntiles), dim3 (SK_NTHREADS), 0, dom->stream, A);
  else if (A.op == 2) {                              /* relax with the face weights of every cell */
    static bool attr = false;
    if (!attr) {
      GFSHIP_HIP (hipFuncSetAttribute ((const void *) relax_skew_loop_kernel<true, 2>,
				       hipFuncAttributeMaxDynamicSharedMemorySize, SK_WBYTES));
      attr = true;
    }
    hipLaunchKernelGGL ((relax_skew_loop_kernel<true, 2>), dim3 (ntiles), dim3 (SK_NTHREADS + 64), SK_WBYTES,
			dom->stream, A);
  }
  else if (dom->wave_loop && nrelax >= 2 && !has_dia)     /* one compute wave per tile */
    hipLaunchKernelGGL (relax_wave_loop_kernel, dim3 (ntiles), dim3 (WV_NTHREADS), 0, dom->stream, A);
  else if (has_dia)
    hipLaunchKernelGGL ((relax_skew_loop_kernel<true, 0>), dim3 (ntiles), dim3 (SK_NTHREADS), 0, dom->stream, A);
  else
    hipLaunchKernelGGL ((relax_skew_loop_kernel<false, 0>), dim3 (ntiles), dim3 (SK_NTHREADS), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  if (ms) {
    GFSHIP_HIP (hipEventRecord (dom->ev1, dom->stream));
    GFSHIP_HIP (hipEventSynchronize (dom->ev1));
    GFSHIP_HIP (hipEventElapsedTime (ms, dom->ev0, dom->ev1));
  }
  if (nrelax >= 2) {
    int block = L.n >= 256 ? 256 : L.n >= 128 ? 128 : 64;
    hipLaunchKernelGGL (skew_loop_ghosts_kernel, dim3 ((L.n + block - 1)/block, L.n, 4), dim3 (block),
			0, dom->stream, A);
    GFSHIP_HIP (hipGetLastError ());
  }
  if (nrelax < 2)
    return GFSHIP_OK;
 dump_stats:
  if (A.stats && ms) {
    (void) hipStreamSynchronize (dom->stream);
    std::vector<u64> h ((size_t) ntiles*SK_MAXF*2);
    (void) hipMemcpy (h.data (), A.stats, h.size ()*sizeof (u64), hipMemcpyDeviceToHost);
    u64 t0 = ~0ull;
    for (int q = 0; q < ntiles; q++) if (h[2*(size_t) q*SK_MAXF] < t0) t0 = h[2*(size_t) q*SK_MAXF];
    for (int Q = 0; Q < S->ntj; Q++)
      for (int P = 0; P < S->ntj; P++)
	if ((P == Q && (P % 5 == 0 || P == S->ntj - 1)) || (Q == 0 && P == S->ntj - 1)) {
	  fprintf (stderr, "tile (%2d,%2d)", P, Q);
	  for (unsigned w = 0; w < nrelax; w++) {
	    size_t o = 2*((size_t) (P + S->ntj*Q)*SK_MAXF + w);
	    fprintf (stderr, "  s%u %7.1f-%7.1f", w, (h[o] - t0)/100., (h[o + 1] - t0)/100.);
	  }
	  fprintf (stderr, " us\n");
	}
  }
  return GFSHIP_OK;
}

} // namespace gfship
