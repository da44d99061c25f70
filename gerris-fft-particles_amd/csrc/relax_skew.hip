// relax_skew.hip -- exact-order Gauss-Seidel sweep as ONE pipelined launch per sweep (3-D).
//
// The reference's in-place sweep (src/poisson.c:507-530, tree pre-order of src/ftt.c:837-852) is
// a lexicographic sweep in the oriented coordinates I = i - 1, J = n - j, K = n - k: cell
// (I,J,K) needs the NEW values of (I-1,J,K), (I,J-1,K), (I,J,K-1) and the OLD values of the +1
// neighbours.  Any schedule that respects these three dependencies gives the same bits.
//
// Schedule.  The (J,K) plane is cut into 16 x 16 tiles; a tile is a 256-thread workgroup and
// every thread owns one full line along I.  Thread (a,b) of a tile works on I = t - a - b at
// step t, so at one step the workgroup is on one hyperplane of its tile, and the values a thread
// needs from threads (a-1,b) and (a,b-1) were produced at step t - 1 (exchanged through LDS,
// one barrier per step).  Tiles run concurrently, each lagging its (J-1) and (K-1) neighbour
// tiles: the new values of the two outgoing edges of a tile are handed to the next tiles
// through small buffers of 8-byte granules pre-filled with a sentinel and written/polled with
// agent-scope (sc1) accesses, so a consumer simply waits for the value it needs (the data is
// the flag; cdna_hip_programming.md Guideline 16, form R2).  Tiles are claimed through a ticket
// in anti-diagonal order, so a waiting tile only ever waits on tiles claimed before it:
// no residency assumption, no deadlock; every spin is bounded.
//
// Layout.  During a relax loop the level lives in a tile-skewed ("hyperplane-major") copy:
// element (I,a,b) of tile T sits at  T*RT*256 + (I + a + b)*256 + (a + 16 b),  RT = n + 30, so
// that at step t every thread reads/writes row t (or t + 1) of its tile: fully coalesced
// streaming with register prefetch, no LDS staging.  The natural array keeps the ghost layer;
// cells adjacent to the box sides are mirrored into it by the sweep so that the BC kernel works
// unchanged between sweeps.
#include "relax_skew.hpp"
#include <cstdlib>

namespace gfship {

// The main loop is written so that the compiler can count outstanding loads exactly
// (s_waitcnt vmcnt(N), never vmcnt(0)): straight-line unrolled body, every lane issues the same
// loads at every step from addresses that are always valid (padded layout), no load sits inside
// a data-dependent loop, and the only divergent blocks hold stores.
//
// Per step a lane loads two values (its own old value one cell ahead and the right-hand side);
// everything it needs from other lines goes through two LDS grids of 17 x 17 doubles:
//   X  new values of the previous step  (column 0 / row 0: values handed over by the (J-1) /
//      (K-1) tiles, or box-side ghosts)
//   Y  old values one cell ahead        (column 16 / row 16: first line of the (J+1) / (K+1)
//      tiles, or box-side ghosts)
// The four halo strips are streamed by the 64 lanes of wave 0 (16 lanes each), which also wait
// for hand-off granules that are still the sentinel (rare slow path).
template <bool HAS_DIA>
__global__ void __launch_bounds__(SK_NL)
relax_skew_kernel (SkewArgs A)
{
  constexpr int XS = SK_T + 1;               // row pitch of the LDS grids
  __shared__ double X[2][XS*XS];
  __shared__ double Y[2][XS*XS];
  __shared__ unsigned s_tile;

  const int tid = threadIdx.x;
  const int a = tid & (SK_T - 1), b = tid >> 4;
  const int n = A.L.n;
  const int ntj = A.ntj;
  const long tstride = (long) (A.RT + 2*SK_FP)*SK_NL;  // doubles per tile incl. padding rows
  const long hstride = (long) SK_HROWS (n)*SK_T;       // granules per tile incl. padding rows

  if (tid == 0)
    s_tile = A.order[atomicAdd (A.ticket, 1u)];
  __syncthreads ();
  const int tile = s_tile;
  u64 st_spins = 0, st_slow = 0;
  const u64 st_start = A.stats ? __builtin_amdgcn_s_memrealtime () : 0;
  const int P = tile % ntj, Q = tile / ntj;   // tile coordinates along J and K
  const int j = n - (SK_T*P + a), k = n - (SK_T*Q + b);   // natural coordinates of the line
  const int s = a + b;

  double * ut = A.us + tile*tstride + SK_FP*SK_NL;

  // ---- own streams: value for step t at base + t*256 ----
  const double * qR = ut + SK_NL + tid;                          // own old value at I + 1
  const double * qRhs = A.rs + tile*tstride + SK_FP*SK_NL + tid;
  const double * qDia = HAS_DIA ? A.ds + tile*tstride + SK_FP*SK_NL + tid : nullptr;

  // ---- halo streams of wave 0: strip g = tid / 16, position m = tid % 16 ----
  const bool loader = __builtin_amdgcn_readfirstlane (tid >> 6) == 0;
  const int g = tid >> 4, m = tid & 15;
  const u64 * qH = A.dummy;     // value for step t at qH + t*hs
  int hs = 0;
  bool handoff = false;         // granule written by another tile in this launch
  int xy_halo = 0;              // LDS index of the halo entry this lane fills
  if (loader) {
    const int jm = n - (SK_T*P + m), km = n - (SK_T*Q + m);
    switch (g) {
    case 0: // new values of line (-1, b = m): tile (P-1,Q) hand-off row I + b = t; box side j = n + 1
      if (P > 0) { qH = A.hbJ + (tile - 1)*hstride + m; hs = SK_T; handoff = true; }
      else       { qH = (const u64 *) (A.un + A.L.idx (1 - m, n + 1, km)); hs = 1; }
      xy_halo = 0 + XS*(m + 1);
      break;
    case 1: // new values of line (a = m, -1): tile (P,Q-1) hand-off; box side k = n + 1
      if (Q > 0) { qH = A.hbK + (tile - ntj)*hstride + m; hs = SK_T; handoff = true; }
      else       { qH = (const u64 *) (A.un + A.L.idx (1 - m, jm, n + 1)); hs = 1; }
      xy_halo = (m + 1) + XS*0;
      break;
    case 2: // old values of line (16, b = m) = line (0,m) of tile (P+1,Q), row I + m = t - 15;
	    // box side j = 0
      if (P + 1 < ntj) { qH = (const u64 *) (ut + tstride - (long) (SK_T - 1)*SK_NL + SK_T*m); hs = SK_NL; }
      else             { qH = (const u64 *) (A.un + A.L.idx (1 - (SK_T - 1 + m), 0, km)); hs = 1; }
      xy_halo = SK_T + XS*m;
      break;
    default: // old values of line (a = m, 16) = line (m,0) of tile (P,Q+1), row t - 15; box side k = 0
      if (Q + 1 < ntj) { qH = (const u64 *) (ut + ntj*tstride - (long) (SK_T - 1)*SK_NL + m); hs = SK_NL; }
      else             { qH = (const u64 *) (A.un + A.L.idx (1 - (m + SK_T - 1), jm, 0)); hs = 1; }
      xy_halo = m + XS*SK_T;
    }
  }
  bool failed = false;          // a bounded wait gave up: stop waiting, the host reports it
  double * const halo_dst0 = (g < 2 ? &X[0][0] : &Y[0][0]) + xy_halo;
  const u64 * const qH0 = qH;

  // ghost cells at the two ends of the line (frozen during the sweep)
  const double ghostL = A.un[A.L.idx (0, j, k)];
  const double ghostR = A.un[A.L.idx (n + 1, j, k)];

  // store pointers: own row t; hand-off row t - 15 of this tile's buffers; natural mirror
  double * wU = ut + tid;
  const bool hasJ = (a == SK_T - 1 && P + 1 < ntj), hasK = (b == SK_T - 1 && Q + 1 < ntj);
  u64 * wJ = A.hbJ + tile*hstride + b - (long) (SK_T - 1)*SK_T;   // dereferenced only if hasJ
  u64 * wK = A.hbK + tile*hstride + a - (long) (SK_T - 1)*SK_T;   // dereferenced only if hasK
  const bool side_jk = (j == 1 || j == n || k == 1 || k == n);
  double * nat = A.un + A.L.idx (1 - s, j, k);   // natural address of I at step t: nat[t]

  // LDS indices
  const int iOwnX = (a + 1) + XS*(b + 1), iT = a + XS*(b + 1), iF = (a + 1) + XS*b;
  const int iOwnY = a + XS*b, iBo = (a + 1) + XS*b, iBk = a + XS*(b + 1);

  // steps, rounded up to the unroll factor (the extra steps find every lane inactive) so that
  // the unrolled body is one straight-line block and load counting stays exact
  const int T = (n + SK_PAD + SK_D - 1)/SK_D*SK_D;
  // register prefetch rings: slot q holds the loads for steps t = q (mod SK_D)
  double pR[SK_D], pRhs[SK_D], pDia[SK_D], pH[SK_DH];

#define SK_PREFETCH(q_)							\
  do {									\
    pR[q_]   = *qR;   qR += SK_NL;					\
    pRhs[q_] = *qRhs; qRhs += SK_NL;					\
    if (HAS_DIA) { pDia[q_] = *qDia; qDia += SK_NL; }			\
  } while (0)
#define SK_PREFETCH_HALO(q_)						\
  do {									\
    if (loader) {							\
      pH[q_] = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs; \
    }									\
  } while (0)

  // wave 0: put the halo value of step t_ (held in slot q_) into LDS buffer t_ & 1
#define SK_HALO(t_, q_)							\
  do {									\
    if (loader) {							\
      double hv = pH[q_];						\
      /* a granule is awaited only for the steps at which the consuming line is active */ \
      bool w = handoff && !failed && (unsigned) ((t_) - m) < (unsigned) n && \
	(u64) __double_as_longlong (hv) == SK_SENTINEL;			\
      if (__builtin_expect (__any (w), 0)) {				\
	unsigned spins = 0;						\
	st_slow++;							\
	while (__any (w)) {						\
	  st_spins++;							\
	  __builtin_amdgcn_s_sleep (2);					\
	  if (w) {							\
	    hv = __longlong_as_double ((long long) load_sc1 (qH0 + (long) (t_)*hs)); \
	    w = (u64) __double_as_longlong (hv) == SK_SENTINEL;		\
	  }								\
	  if (++spins > (1u << 16)) { *A.err = 1; failed = true; break; } \
	}								\
      }									\
      halo_dst0[((t_) & 1)*(XS*XS)] = hv;				\
    }									\
  } while (0)

  // the halo stream runs one step ahead of the others: its value for step t + 1 is put into
  // LDS during step t, so halo slot q holds the value of step t + 1 for t = q (mod SK_DH)
  double h0 = 0.;
  if (loader) { h0 = __longlong_as_double ((long long) load_sc1 (qH)); qH += hs; }
#pragma unroll
  for (int q = 0; q < SK_D; q++)
    SK_PREFETCH (q);
#pragma unroll
  for (int q = 0; q < SK_DH; q++)
    SK_PREFETCH_HALO (q);

  // LDS state for step 0: old values one cell ahead and the halos of step 0
  for (int q = tid; q < 2*XS*XS; q += SK_NL) {
    (&X[0][0])[q] = 0.;
    (&Y[0][0])[q] = 0.;
  }
  __syncthreads ();
  Y[0][iOwnY] = pR[0];
  {
    double keep = pH[0];
    pH[0] = h0;
    SK_HALO (0, 0);
    pH[0] = keep;
  }
  __syncthreads ();

  double prev = ghostL;

  for (int t0 = 0; t0 < T; t0 += SK_D) {
#pragma unroll
    for (int q = 0; q < SK_D; q++) {
      const int t = t0 + q;
      const int I = t - s;
      const bool act = I >= 0 && I < n;
      const int B = t & 1;
      // values of the other lines: new ones of step t - 1, old ones one cell ahead
      const double Tn = X[B][iT], Fn = X[B][iF], Bo = Y[B][iBo], Bk = Y[B][iBk];
      // relax, src/poisson.c:507-530, unit weights, d = 0..5 = right,left,top,bottom,front,back
      const double Rv = (I + 1 < n) ? pR[q] : ghostR;
      double aa = HAS_DIA ? pDia[q] : 0., bb = 0.;
      aa += 1.; bb += 1.*Rv;        // right  (+x, old)
      aa += 1.; bb += 1.*prev;      // left   (-x, new)
      aa += 1.; bb += 1.*Tn;        // top    (+y = J-1, new)
      aa += 1.; bb += 1.*Bo;        // bottom (-y = J+1, old)
      aa += 1.; bb += 1.*Fn;        // front  (+z = K-1, new)
      aa += 1.; bb += 1.*Bk;        // back   (-z = K+1, old)
      const double v = HAS_DIA ? (aa != 0. ? (bb - pRhs[q])/aa : 0.) : divide_by_6 (bb - pRhs[q]);
      prev = act ? v : prev;
      // publish for step t + 1
      X[B ^ 1][iOwnX] = v;
      Y[B ^ 1][iOwnY] = pR[(q + 1) % SK_D];
      // wave 0: halo of step t + 1 into LDS; then refill slot q (steps t + SK_D, halo t + 1 + SK_D)
      SK_HALO (t + 1, q % SK_DH);
      SK_PREFETCH_HALO (q % SK_DH);
      SK_PREFETCH (q);
      // own row of the skewed copy: stored unconditionally (for an inactive lane the slot is
      // padding that nothing reads), which keeps the common path free of branches
      *wU = v;
      wU += SK_NL;
      if (act && (hasJ || hasK || side_jk || I == 0 || I == n - 1)) {
	if (hasJ) store_sc1 (wJ, (u64) __double_as_longlong (v));
	if (hasK) store_sc1 (wK, (u64) __double_as_longlong (v));
	if (side_jk || I == 0 || I == n - 1)
	  nat[t] = v;
      }
      wJ += SK_T;
      wK += SK_T;
      // workgroup barrier that drains LDS traffic only: __syncthreads() would also wait for the
      // prefetch loads just issued (s_waitcnt vmcnt(0)) and serialise every step on HBM latency
      asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
#undef SK_PREFETCH
#undef SK_PREFETCH_HALO
#undef SK_HALO
  if (A.stats && tid == 0) {
    u64 * st = A.stats + 4*tile;
    st[0] = st_start; st[1] = __builtin_amdgcn_s_memrealtime (); st[2] = st_spins; st[3] = st_slow;
  }
}

// ---------------------------------------------------------------------------------------------
// natural <-> skewed copies through an LDS transpose.  One block = one tile, one b (a K plane of
// 16 lines), one chunk of 16 cells along I: reads and writes are 128-byte segments on both sides.
// Element (I,a,b) of the tile <-> skewed row I + a + b, column a + 16 b.
// ---------------------------------------------------------------------------------------------
struct PackArgs {
  Layout L;
  int ntj, RT;
  const double * src[3];
  double * dst[3];
  int narr;
  double * add;            // unpack: add the values to this natural array instead of storing them
};

__global__ void __launch_bounds__(SK_NL)
skew_pack_kernel (PackArgs A)
{
  __shared__ double tile_[3][SK_T][SK_T + 1];
  const int tid = threadIdx.x;
  const int lo = tid & (SK_T - 1), hi = tid >> 4;
  const int tile = blockIdx.z;
  const int b = blockIdx.y;
  const int I0 = blockIdx.x*SK_T;
  const int P = tile % A.ntj, Q = tile / A.ntj;
  const int n = A.L.n;
  const int k = n - (SK_T*Q + b);
  const long tbase = (long) tile*(A.RT + 2*SK_FP)*SK_NL + SK_FP*SK_NL;
  // read: hi = a, lo = cell along I (contiguous in memory)
  {
    const int j = n - (SK_T*P + hi);
    long nidx = A.L.idx (I0 + lo + 1, j, k);
    for (int q = 0; q < A.narr; q++)
      tile_[q][hi][lo] = A.src[q][nidx];
  }
  __syncthreads ();
  // write: lo = a (contiguous in the skewed row), hi selects the diagonal m = I - I0 + a
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int m = hi + SK_T*pass;            // 0 .. 30
    const int di = m - lo;                   // I - I0
    if (m <= 2*SK_T - 2 && di >= 0 && di < SK_T) {
      long sidx = tbase + (long) (I0 + m + b)*SK_NL + lo + SK_T*b;
      for (int q = 0; q < A.narr; q++)
	A.dst[q][sidx] = tile_[q][lo][di];
    }
  }
}

__global__ void __launch_bounds__(SK_NL)
skew_unpack_kernel (PackArgs A)
{
  __shared__ double tile_[SK_T][SK_T + 1];
  const int tid = threadIdx.x;
  const int lo = tid & (SK_T - 1), hi = tid >> 4;
  const int tile = blockIdx.z;
  const int b = blockIdx.y;
  const int I0 = blockIdx.x*SK_T;
  const int P = tile % A.ntj, Q = tile / A.ntj;
  const int n = A.L.n;
  const int k = n - (SK_T*Q + b);
  const long tbase = (long) tile*(A.RT + 2*SK_FP)*SK_NL + SK_FP*SK_NL;
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int m = hi + SK_T*pass;
    const int di = m - lo;
    if (m <= 2*SK_T - 2 && di >= 0 && di < SK_T)
      tile_[lo][di] = A.src[0][tbase + (long) (I0 + m + b)*SK_NL + lo + SK_T*b];
  }
  __syncthreads ();
  const int j = n - (SK_T*P + hi);
  const long c = A.L.idx (I0 + lo + 1, j, k);
  if (A.add)
    A.add[c] += tile_[hi][lo];     /* correct (src/poisson.c:998-1003) fused into the unpack */
  else
    A.dst[0][c] = tile_[hi][lo];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
bool skew_supported (const gfship_domain * dom, int level)
{
  if (dom->dim != 3) return false;
  int n = dom->lay[level].n;
  if (n < 2*SK_T) return false;          /* smaller levels run the LDS relax loop */
  /* GFSHIP_SIDE_EXTERNAL sides are fine: with the reference's `overlap = 0` order their ghost
     cells are frozen during a sweep like any other ghost cell */
  return true;
}

// the "a bounded wait gave up" flag of a level lives in host memory mapped on the device (written
// by a kernel only when it happens): the host checks it without a copy
static unsigned * skew_err_word (gfship_domain * dom, int level)
{
  return (unsigned *) (dom->h_pinned + 32) + level;
}

static int skew_plan (gfship_domain * dom, int level, SkewPlan ** out)
{
  SkewPlan & S = dom->skew[level];
  if (!S.us) {
    const Layout & L = dom->lay[level];
    S.ntj = L.n/SK_T;
    S.RT = L.n + SK_PAD;
    int ntiles = S.ntj*S.ntj;
    size_t doubles = (size_t) ntiles*(S.RT + 2*SK_FP)*SK_NL + 64;  /* padding rows per tile */
    S.hb_words = (size_t) ntiles*SK_HROWS (L.n)*SK_T;   /* rows n + 15 used, rest padding */
    GFSHIP_HIP (hipMalloc ((void **) &S.us, doubles*sizeof (double)));
    GFSHIP_HIP (hipMalloc ((void **) &S.rs, doubles*sizeof (double)));
    GFSHIP_HIP (hipMalloc ((void **) &S.ds, doubles*sizeof (double)));
    GFSHIP_HIP (hipMalloc ((void **) &S.hb, 2*S.hb_words*sizeof (u64)));
    GFSHIP_HIP (hipMalloc ((void **) &S.ctl, 64));
    if (getenv ("GFSHIP_SKEW_STATS"))
      GFSHIP_HIP (hipMalloc ((void **) &S.stats, (size_t) ntiles*4*sizeof (u64)));
    // ticket -> tile, anti-diagonal major: a tile only depends on tiles of earlier diagonals
    std::vector<unsigned short> order;
    for (int d = 0; d <= 2*(S.ntj - 1); d++)
      for (int Q = 0; Q < S.ntj; Q++) {
	int P = d - Q;
	if (P >= 0 && P < S.ntj)
	  order.push_back ((unsigned short) (P + S.ntj*Q));
      }
    GFSHIP_HIP (hipMalloc ((void **) &S.order, order.size ()*sizeof (unsigned short)));
    GFSHIP_HIP (hipMemcpy (S.order, order.data (), order.size ()*sizeof (unsigned short),
			   hipMemcpyHostToDevice));
    if (ntiles >= 8 && S.ntj % 4 == 0) {
      // blocks of (ntj/2) x (ntj/4) tiles, one per XCD, each in anti-diagonal order
      int bw = S.ntj/2, bh = S.ntj/4;
      std::vector<unsigned short> xo;
      for (int x = 0; x < 8; x++)
	for (unsigned short t : order) {
	  int P = t % S.ntj, Q = t / S.ntj;
	  if ((P/bw) + 2*(Q/bh) == x) xo.push_back (t);
	}
      GFSHIP_HIP (hipMalloc ((void **) &S.xorder, xo.size ()*sizeof (unsigned short)));
      GFSHIP_HIP (hipMemcpy (S.xorder, xo.data (), xo.size ()*sizeof (unsigned short),
			     hipMemcpyHostToDevice));
    }
    GFSHIP_HIP (hipMemsetAsync (S.ctl, 0, 64, dom->stream));
  }
  *out = &S;
  return GFSHIP_OK;
}

void skew_free (gfship_domain * dom)
{
  for (int l = 0; l <= GFSHIP_MAXLEVEL; l++) {
    SkewPlan & S = dom->skew[l];
    if (S.us) (void) hipFree (S.us);
    if (S.rs) (void) hipFree (S.rs);
    if (S.ds) (void) hipFree (S.ds);
    for (int d = 0; d < 6; d++) if (S.ws[d]) (void) hipFree (S.ws[d]);
    if (S.hb) (void) hipFree (S.hb);
    if (S.hbf) (void) hipFree (S.hbf);
    if (S.arm_cum) (void) hipFree (S.arm_cum);
    if (S.arm_ev) (void) hipEventDestroy (S.arm_ev);
    if (S.stats_loop) (void) hipFree (S.stats_loop);
    if (S.ctl) (void) hipFree (S.ctl);
    if (S.stats) (void) hipFree (S.stats);
    if (S.order) (void) hipFree (S.order);
    if (S.xorder) (void) hipFree (S.xorder);
    S = SkewPlan ();
  }
}

static int skew_pack (gfship_domain * dom, int level, SkewPlan * S, const double * u,
		      const double * rhs, const double * dia, const double * coarse = nullptr)
{
  if (patch_level (dom, level))
    return patch_pack (dom, level, S, u, rhs, dia, coarse);
  PackArgs A;
  A.add = nullptr;
  A.L = dom->lay[level]; A.ntj = S->ntj; A.RT = S->RT;
  A.narr = 0;
  A.src[A.narr] = u;   A.dst[A.narr++] = S->us;
  A.src[A.narr] = rhs; A.dst[A.narr++] = S->rs;
  if (dia) { A.src[A.narr] = dia; A.dst[A.narr++] = S->ds; }
  dim3 grid (A.L.n/SK_T, SK_T, S->ntj*S->ntj);
  hipLaunchKernelGGL (skew_pack_kernel, grid, dim3 (SK_NL), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// the six face weights of a level (gfs_poisson_coefficients with alpha) into the layout of the six-wave
// kernel: once per recomputation of the weights (weights_stamp), whatever the number of sweeps
static int skew_pack_weights (gfship_domain * dom, int level, SkewPlan * S, const RelaxOp * op)
{
  if (S->ws[0] && S->ws_stamp == dom->weights_stamp)
    return GFSHIP_OK;
  const size_t doubles = (size_t) S->ntj*S->ntj*(S->RT + 2*SK_FP)*SK_NL + 64;
  for (int d = 0; d < 6; d++)
    if (!S->ws[d]) {
      GFSHIP_HIP (hipMalloc ((void **) &S->ws[d], doubles*sizeof (double)));
      GFSHIP_HIP (hipMemsetAsync (S->ws[d], 0, doubles*sizeof (double), dom->stream));
    }
  for (int h = 0; h < 2; h++) {
    PackArgs A;
    A.add = nullptr;
    A.L = dom->lay[level]; A.ntj = S->ntj; A.RT = S->RT;
    A.narr = 3;
    for (int q = 0; q < 3; q++) { A.src[q] = op->wf[3*h + q]; A.dst[q] = S->ws[3*h + q]; }
    dim3 grid (A.L.n/SK_T, SK_T, S->ntj*S->ntj);
    hipLaunchKernelGGL (skew_pack_kernel, grid, dim3 (SK_NL), 0, dom->stream, A);
    GFSHIP_HIP (hipGetLastError ());
  }
  S->ws_stamp = dom->weights_stamp;
  return GFSHIP_OK;
}

static int skew_unpack (gfship_domain * dom, int level, SkewPlan * S, double * u,
			double * add_into = nullptr)
{
  if (patch_level (dom, level))
    return patch_unpack (dom, level, S, u, add_into);
  PackArgs A;
  A.L = dom->lay[level]; A.ntj = S->ntj; A.RT = S->RT;
  A.narr = 1;
  A.src[0] = S->us; A.dst[0] = u; A.add = add_into;
  dim3 grid (A.L.n/SK_T, SK_T, S->ntj*S->ntj);
  hipLaunchKernelGGL (skew_unpack_kernel, grid, dim3 (SK_NL), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

static int skew_launch (gfship_domain * dom, int level, SkewPlan * S, double * u_nat, bool has_dia)
{
  SkewArgs A;
  A.L = dom->lay[level]; A.ntj = S->ntj; A.RT = S->RT;
  A.us = S->us; A.rs = S->rs; A.ds = has_dia ? S->ds : nullptr;
  A.un = u_nat;
  A.hbJ = (u64 *) S->hb; A.hbK = (u64 *) S->hb + S->hb_words;
  A.order = S->order;
  A.ticket = (unsigned *) S->ctl;
  A.err = skew_err_word (dom, level);
  A.dummy = (const u64 *) S->ctl + 2;
  A.stats = S->stats;
  int ntiles = S->ntj*S->ntj;
  if (has_dia)
    hipLaunchKernelGGL (relax_skew_kernel<true>, dim3 (ntiles), dim3 (SK_NL), 0, dom->stream, A);
  else
    hipLaunchKernelGGL (relax_skew_kernel<false>, dim3 (ntiles), dim3 (SK_NL), 0, dom->stream, A);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

static int skew_sweep (gfship_domain * dom, int level, SkewPlan * S, double * u_nat, bool has_dia,
		       const RelaxOp * op = nullptr)
{
  /* the six-wave kernel of relax_skew_loop.hip also runs a single sweep (any sides); the
     four-wave kernel below is kept as an independent implementation (GFSHIP_SKEW_OLD=1) */
  if (!dom->skew_old || (op && op->kind))
    return skew_loop_run (dom, level, S, u_nat, has_dia, 1, nullptr, nullptr, op);
  // ticket = 0 (err is sticky), hand-off granules = sentinel
  GFSHIP_HIP (hipMemsetAsync (S->ctl, 0, sizeof (unsigned), dom->stream));
  if (S->ntj > 1)
    GFSHIP_HIP (hipMemsetAsync (S->hb, 0xFF, 2*S->hb_words*sizeof (u64), dom->stream));
  return skew_launch (dom, level, S, u_nat, has_dia);
}

// the whole relax_loop (src/poisson.c:1070-1089) of one level in the skewed layout; for
// nrelax = 1 and bc = false it is a single gfs_relax sweep
// Once per level: a trial run of the fused loop on zeros.  Its tiles wait on each other, so all of
// them must really be resident at the same time; the occupancy query says they are, but if the
// device does not deliver (CUs in use by someone else ...) the bounded waits time out and report it
// -- then this domain keeps to one launch per sweep.
static int skew_loop_trial (gfship_domain * dom, int level, SkewPlan * S, unsigned nrelax, bool bc)
{
  int r;
  if (!dom->no_fused_loop && !S->loop_checked && skew_loop_supported (dom, level, nrelax, bc)) {
    S->loop_checked = true;
    const Layout & L = dom->lay[level];
    size_t doubles = (size_t) S->ntj*S->ntj*(S->RT + 2*SK_FP)*SK_NL;
    double * zero_nat = nullptr;
    GFSHIP_HIP (hipMalloc ((void **) &zero_nat, L.total*sizeof (double)));
    GFSHIP_HIP (hipMemsetAsync (zero_nat, 0, L.total*sizeof (double), dom->stream));
    GFSHIP_HIP (hipMemsetAsync (S->us, 0, doubles*sizeof (double), dom->stream));
    GFSHIP_HIP (hipMemsetAsync (S->rs, 0, doubles*sizeof (double), dom->stream));
    r = skew_loop_run (dom, level, S, zero_nat, false, 2);
    if (r == GFSHIP_OK)
      GFSHIP_HIP (hipStreamSynchronize (dom->stream));
    (void) hipFree (zero_nat);
    if (r != GFSHIP_OK) return r;
    if (*skew_err_word (dom, level)) {
      dom->no_fused_loop = true;
      *skew_err_word (dom, level) = 0;
    }
  }
  return GFSHIP_OK;
}

bool prolongation_fused (gfship_domain * dom, unsigned dimension, int level, unsigned nrelax)
{
  if (dom->no_fused_prolongation || dom->weighted || level < 1) return false;
  if (dom->relax_mode != GFSHIP_RELAX_EXACT || dimension != 3 || dom->force_hyperplane) return false;
  if (dom->has_external && dom->overlap && nrelax > 1) return false;
  return skew_supported (dom, level) && patch_level (dom, level);
}

int launch_relax_loop_skew (gfship_domain * dom, int level, Field * dp, Field * ubc,
			    const double * rhs, const double * dia, bool dia_zero,
			    unsigned nrelax, bool bc, double * correct_into, const double * prolong_from,
			    const RelaxOp * op)
{
  SkewPlan * S;
  int r;
  if ((r = skew_plan (dom, level, &S))) return r;
  double * u = dp->lev[level];
  if ((r = skew_loop_trial (dom, level, S, nrelax, bc))) return r;
  if (op && op->kind == 2) {
    if ((r = skew_pack_weights (dom, level, S, op))) return r;
    dia_zero = false;           /* the weighted cell update always reads dia */
  }
  /* get_from_above fused into the copy into the skewed layout: the natural array gets the cells
     along the box sides only (what the BC application reads), then the BC, then the loop */
  const bool fusedp = prolong_from != nullptr && bc && patch_level (dom, level);
  /* the residual was copied into the skewed layout when it was restricted (launch_restrict_pack) */
  const double * rhs_pack = S->rs_ready && patch_level (dom, level) ? nullptr : rhs;
  S->rs_ready = false;
  if (prolong_from && !fusedp)
    if ((r = launch_prolongate (dom, level - 1, prolong_from, u))) return r;
  if (fusedp && (r = skew_pack (dom, level, S, u, rhs_pack, dia_zero ? nullptr : dia, prolong_from))) return r;
  if (bc && (r = launch_bc (dom, ubc, dp, level, 1))) return r;
  if (!fusedp && (r = skew_pack (dom, level, S, u, rhs_pack, dia_zero ? nullptr : dia))) return r;
  if (!dom->no_fused_loop && skew_loop_supported (dom, level, nrelax, bc)) {
    /* the sweeps of the loop pipelined in one launch (relax_skew_loop.hip) */
    if ((r = skew_loop_run (dom, level, S, u, !dia_zero, nrelax, nullptr, ubc, op))) return r;
    return skew_unpack (dom, level, S, u, correct_into);
  }
  for (unsigned q = 0; q < nrelax; q++) {
    if ((r = skew_sweep (dom, level, S, u, !dia_zero, op))) return r;
    if (bc && q + 1 < nrelax && (r = launch_bc (dom, ubc, dp, level, 1))) return r;
  }
  return skew_unpack (dom, level, S, u, correct_into);
}

int launch_restrict_pack (gfship_domain * dom, unsigned dimension, int level, Field * res, bool coarse_fused)
{
  SkewPlan * S, * Sc = nullptr;
  int r;
  if ((r = skew_plan (dom, level, &S))) return r;
  /* the trial run of the fused loop overwrites the skewed arrays: before they are filled */
  if ((r = skew_loop_trial (dom, level, S, 4, true))) return r;
  if (coarse_fused) {
    if ((r = skew_plan (dom, level - 1, &Sc))) return r;
    if ((r = skew_loop_trial (dom, level - 1, Sc, 4, true))) return r;
  }
  res->zero[level - 1] = false;
  if ((r = patch_restrict_pack (dom, level, S, res->lev[level], res->lev[level - 1], Sc, dimension))) return r;
  S->rs_ready = true;
  if (Sc) Sc->rs_ready = true;
  return GFSHIP_OK;
}

// debug: print per-tile timing of the last sweep of `level` (GFSHIP_SKEW_STATS=1)
void skew_dump_stats (gfship_domain * dom, int level)
{
  SkewPlan & S = dom->skew[level];
  if (!S.stats) return;
  (void) hipStreamSynchronize (dom->stream);
  int nt = S.ntj*S.ntj;
  std::vector<u64> h (4*nt);
  (void) hipMemcpy (h.data (), S.stats, h.size ()*sizeof (u64), hipMemcpyDeviceToHost);
  u64 t0 = ~0ull;
  for (int q = 0; q < nt; q++) if (h[4*q] < t0) t0 = h[4*q];
  for (int Q = 0; Q < S.ntj; Q++)
    for (int P = 0; P < S.ntj; P++) {
      int q = P + S.ntj*Q;
      if (P == Q || P == 0 || Q == 0 || P == S.ntj - 1)
	fprintf (stderr, "tile (%2d,%2d) start %8.2f us  dur %8.2f us  spins %6llu slow %5llu\n", P, Q,
		 (h[4*q] - t0)/100., (h[4*q + 1] - h[4*q])/100., h[4*q + 2], h[4*q + 3]);
    }
}

// time `reps` sweeps of the pipelined kernel alone (HIP events on the domain's stream)
int skew_time_sweeps (gfship_domain * dom, int level, Field * u, const double * rhs,
		      const double * dia, bool dia_zero, int reps, double * ms_per_sweep)
{
  SkewPlan * S;
  int r;
  if ((r = skew_plan (dom, level, &S))) return r;
  if ((r = skew_pack (dom, level, S, u->lev[level], rhs, dia_zero ? nullptr : dia))) return r;
  if ((r = skew_sweep (dom, level, S, u->lev[level], !dia_zero))) return r; /* warm-up */
  float total = 0.f;
  for (int q = 0; q < reps; q++) {
    float ms = 0.f;
    if (!dom->skew_old) {
      if ((r = skew_loop_run (dom, level, S, u->lev[level], !dia_zero, 1, &ms))) return r;
    }
    else {
      // the memsets that re-arm the hand-off buffers are outside the timed region
      GFSHIP_HIP (hipMemsetAsync (S->ctl, 0, sizeof (unsigned), dom->stream));
      if (S->ntj > 1)
	GFSHIP_HIP (hipMemsetAsync (S->hb, 0xFF, 2*S->hb_words*sizeof (u64), dom->stream));
      GFSHIP_HIP (hipEventRecord (dom->ev0, dom->stream));
      if ((r = skew_launch (dom, level, S, u->lev[level], !dia_zero))) return r;
      GFSHIP_HIP (hipEventRecord (dom->ev1, dom->stream));
      GFSHIP_HIP (hipEventSynchronize (dom->ev1));
      GFSHIP_HIP (hipEventElapsedTime (&ms, dom->ev0, dom->ev1));
    }
    total += ms;
  }
  *ms_per_sweep = total/reps;
  return skew_unpack (dom, level, S, u->lev[level]);
}

// time `reps` relax loops (nrelax sweeps, homogeneous BC of u itself between them): the sweep
// kernels alone (ms_per_loop), and -- ms_inclusive -- everything the V-cycle runs on this level between
// the relax loop of the level below and the corrected solution, as poisson_cycle runs it: the
// prolongation of the coarser level straight into the layout of the loop (get_from_above,
// patch_prolong_kernel; the plain copy on the levels of the six-wave kernel), the BC kernel, the
// sweeps, the ghost planes, and the way out of the layout with the correction `u += dp' in it
// (patch_unpack_kernel with add).  The granules are armed before the timed region: in a V-cycle
// that fill runs on the side stream beside the relax loops of the coarser levels (skew_arm_ahead).
// The right-hand side is put into the layout once, outside the timed region (in a V-cycle: by the
// restriction of the residual on the way down, patch_restrict_pack).
int skew_time_loop (gfship_domain * dom, int level, Field * u, const double * rhs,
		    const double * dia, bool dia_zero, unsigned nrelax, int reps,
		    double * ms_per_loop, int * fused, double * ms_inclusive)
{
  SkewPlan * S;
  int r;
  if ((r = skew_plan (dom, level, &S))) return r;
  double * un = u->lev[level];
  bool fuse = !dom->no_fused_loop && skew_loop_supported (dom, level, nrelax, true);
  *fused = fuse;
  double total = 0., total_incl = 0.;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double * target = nullptr;          /* the solution the correction is added to */
  const bool vcycle_like = ms_inclusive && fuse && level >= 1 && patch_level (dom, level);
  if (ms_inclusive) {
    GFSHIP_HIP (hipEventCreate (&e0));
    GFSHIP_HIP (hipEventCreate (&e1));
  }
  if (vcycle_like) {
    GFSHIP_HIP (hipMalloc ((void **) &target, dom->lay[level].total*sizeof (double)));
    GFSHIP_HIP (hipMemsetAsync (target, 0, dom->lay[level].total*sizeof (double), dom->stream));
    /* the rhs (dia) once; u comes from the coarser level in every repetition */
    if ((r = skew_pack (dom, level, S, un, rhs, dia_zero ? nullptr : dia))) return r;
  }
  for (int q = -1; q < reps; q++) {       /* q = -1: warm-up */
    float ms = 0.f;
    if (vcycle_like) {
      /* armed ahead, outside the timed region */
      if (S->hbf) {
        if ((r = skew_arm_ahead (dom, level, nrelax))) return r;
        GFSHIP_HIP (hipStreamSynchronize (dom->side_stream));
      }
      GFSHIP_HIP (hipEventRecord (e0, dom->stream));
      if ((r = skew_pack (dom, level, S, un, nullptr, nullptr, u->lev[level - 1]))) return r;
      if ((r = launch_bc (dom, u, u, level, 1))) return r;
      if ((r = skew_loop_run (dom, level, S, un, !dia_zero, nrelax, &ms, u))) return r;
      if ((r = skew_unpack (dom, level, S, un, target))) return r;
    }
    else {
      /* levels / modes without the fused path: BC kernel, copy in, sweeps, copy out */
      if (e0) GFSHIP_HIP (hipEventRecord (e0, dom->stream));
      if ((r = launch_bc (dom, u, u, level, 1))) return r;
      if ((r = skew_pack (dom, level, S, un, rhs, dia_zero ? nullptr : dia))) return r;
      if (fuse) {
	if ((r = skew_loop_run (dom, level, S, un, !dia_zero, nrelax, &ms, u))) return r;
      }
      else
	for (unsigned w = 0; w < nrelax; w++) {
	  float m1 = 0.f;
	  if (!dom->skew_old) {
	    if ((r = skew_loop_run (dom, level, S, un, !dia_zero, 1, &m1))) return r;
	  }
	  else {
	    GFSHIP_HIP (hipMemsetAsync (S->ctl, 0, sizeof (unsigned), dom->stream));
	    if (S->ntj > 1)
	      GFSHIP_HIP (hipMemsetAsync (S->hb, 0xFF, 2*S->hb_words*sizeof (u64), dom->stream));
	    GFSHIP_HIP (hipEventRecord (dom->ev0, dom->stream));
	    if ((r = skew_launch (dom, level, S, un, !dia_zero))) return r;
	    GFSHIP_HIP (hipEventRecord (dom->ev1, dom->stream));
	    GFSHIP_HIP (hipEventSynchronize (dom->ev1));
	    GFSHIP_HIP (hipEventElapsedTime (&m1, dom->ev0, dom->ev1));
	  }
	  ms += m1;
	  if (w + 1 < nrelax && (r = launch_bc (dom, u, u, level, 1))) return r;
	}
      if ((r = skew_unpack (dom, level, S, un))) return r;
    }
    if (e0) {
      float mi = 0.f;
      GFSHIP_HIP (hipEventRecord (e1, dom->stream));
      GFSHIP_HIP (hipEventSynchronize (e1));
      GFSHIP_HIP (hipEventElapsedTime (&mi, e0, e1));
      if (q >= 0) total_incl += mi;
    }
    if (q >= 0) total += ms;
  }
  *ms_per_loop = total/reps;
  if (ms_inclusive) {
    *ms_inclusive = total_incl/reps;
    (void) hipEventDestroy (e0);
    (void) hipEventDestroy (e1);
  }
  if (target) {
    GFSHIP_HIP (hipStreamSynchronize (dom->stream));
    (void) hipFree (target);
  }
  return GFSHIP_OK;
}

int skew_check_error (gfship_domain * dom)
{
  for (int l = 0; l <= dom->depth; l++)
    if (dom->skew[l].ctl && *skew_err_word (dom, l) != 0) {
      /* the word is not sticky: this solve is lost (its tiles computed on values that never
	 arrived), but the domain keeps working -- from now on with one launch per sweep, whose
	 tiles only ever wait on tiles claimed before them (no residency assumption) */
      *skew_err_word (dom, l) = 0;
      dom->skew[l].armed[0] = dom->skew[l].armed[1] = 0;
      dom->no_fused_loop = true;
      set_error ("relax_skew_kernel: a hand-off wait timed out on level %d (CUs held by another "
		 "process or stream?); this solve failed, the domain falls back to one launch "
		 "per sweep", l);
      return GFSHIP_EHIP;
    }
  return GFSHIP_OK;
}

} // namespace gfship
