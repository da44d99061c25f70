// tree_flow.hpp -- the relax loop of a tree level (src/poisson.c:1070-1089 over the cells of
// T_LEVEL_LEAFS, src/poisson.c:604-632) as a dataflow program of fixed-format micro-operations.
// Included by tree.hip inside its anonymous namespace (after Sweep, Ghost, Sgn6, the K_* codes).
//
// Why: t_relax_nodes interprets the stencil of a cell from a variable-length tape, one thread per
// cell -- a coarse leaf with fine neighbours is a chain of ~100 dependent LDS reads, and the slowest
// thread of a dependency level sets its time (8-10 us per level on the fine levels of the octree
// bench, profiles/r03_tree_flow.txt).  Here the stencil of a cell is cut where its expression tree
// branches (same floating-point operations in the same order):
//   FC     gradient_fine_coarse towards a coarser neighbour (src/fluid.c:283-309 through
//          interpolate_1D1 / interpolate_2D1, :178-245): the pair (nb, na) of one face;
//   CHILD  the same expression seen from the coarse side, one fine cell behind a face of a coarser
//          leaf (src/fluid.c:795-829): the pair of one child;
//   SUM    the sum over the children of that face, in child order (:812-826);
//   CELL   the sums over the faces in direction order and the new value (relax, src/poisson.c:507-557;
//          diffusion_relax, :1455-1484);
//   GHOST  the copy of a ghost cell between two sweeps (homogeneous condition or periodic image).
// Every operation has at most nine inputs at fixed places of a 48-byte record, all independent
// loads, and is scheduled as a node of its own: CHILD one level before its SUM, FC / SUM one level
// before their CELL.  A level holds at most 1024 operations: thread t of the workgroup executes
// operation t of the level and leaves its result in LDS (double-buffered by the parity of the level)
// for the operations of the NEXT level -- that is how the pairs travel, and how a cell value reaches
// the neighbours that follow it immediately.  Values older than one level are read from global memory,
// one level ahead of their use (the workgroup's own stores of earlier levels are ordered before these
// loads by the barrier: same CU, no acknowledgement needed); the record of a level is loaded two levels
// ahead.  What is left on the chain of a level: the LDS reads of its fresh inputs, the arithmetic of
// ONE micro-operation, an LDS write, a barrier.
//
// The schedule keeps the order of the sequential program as loop_plan does (a read follows the write
// it must see, a write follows the reads of the value it replaces, one level at least), so that the
// single image in global memory is right for every load; gfship_tree_host_check runs the plan on the
// host with the kernel's timing of loads and stores and compares with the reference's program.

enum { F_CELL = 0, F_FC = 1, F_CHILD = 2, F_SUM = 3, F_GHOST = 4, F_NOP = 5 };
enum { FK_NONE = 0, FK_SAME = 1, FK_PAIR = 2 };
#define FLOW_LDS    0x40000000      /* input = the result of operation (ref & 511) of one of the last FLOW_PD levels, */
#define FLOW_SLOT(ref) (((ref) & FLOW_LDS) ? ((ref) & 2047) : 0)   /* in slot (level % FLOW_NBUF)*FLOW_WIDTH + operation */
#define FLOW_WIDTH  512             /* operations per level = threads of the workgroup */
#define FLOW_PD     2               /* old values are loaded FLOW_PD levels ahead, records FLOW_PD + 2 */
#define FLOW_NBUF   (FLOW_PD + 1)
#define FLOW_NIN    10
#define FLOW_NCONST 128

struct __attribute__((aligned(16))) FlowRec { unsigned w0; int out_g; int in[FLOW_NIN]; };
// w0: bits 0-2 the kind
//   CELL   3-14 the kind of each face (FK_*), 15-19 the level of the cell, 20: reads its own value
//          in[0] the cell itself, in[1 + d] the neighbour (FK_SAME) or the pair of the face (FK_PAIR)
//   FC / CHILD  3-4 terms of the interpolation, 5-7 / 8-10 cells averaged in term 0 / 1 (0: one
//          value), 11-17 / 18-24 / 25-31 the constants cb (gbi), a0, a1 in the table
//          in[0] the coarse neighbour (the child), in[1 + TS t + k] value k of term t
//   SUM    3-5 children;  in[i] the pair of child i
//   GHOST  11-17 the constant (the sign of the homogeneous condition);  in[0] the image

template <int DIM> struct FlowShape {
  static constexpr int TS = DIM == 3 ? 4 : 2;            /* values of one interpolation term */
  static constexpr int NIN = DIM == 3 ? 9 : 5;           /* inputs a record of this dimension uses */
};

struct FlowPair { double x, y; };

// one micro-operation, arithmetic only.  x[j]: the input j (a value from global memory or the result of an
// operation of the previous level), y[j]: the second number of the pair when input j is a pair; rh: the
// right-hand side of a CELL; ct: the constants
template <int DIM>
__host__ __device__ inline FlowPair flow_eval (unsigned w0, const double * x, const double * y, double rh,
					       const double * ct, double omega, int op, double w)
{
  constexpr int TS = FlowShape<DIM>::TS;
  const int kind = w0 & 7;
  FlowPair o = { 0., 0. };
  if (kind == F_CELL) {
    double ga = 0., gb = 0.;
#pragma unroll
    for (int d = 0; d < 2*DIM; d++) {
      const int fk = (w0 >> (3 + 2*d)) & 3;
      if (fk == FK_SAME) {
	const double na = w, nb = w*x[1 + d];
	ga += na; gb += nb;
      }
      else if (fk == FK_PAIR) {
	ga += y[1 + d]; gb += x[1 + d];
      }
    }
    if (op == 0) {
      double r = 0.;
      if (ga != 0.) {
	if (DIM == 2)
	  r = (1. - omega)*x[0] + omega*(gb - rh)/ga;
	else
	  r = (gb - rh)/ga;
      }
      o.x = r;
    }
    else {      /* diffusion_relax, src/poisson.c:1455-1484 (rhoc = 1) */
      const int l = (w0 >> 15) & 31;
      const double h = 1./(1 << l);
      const double a = 1.*h*h;
      ga = 1. + ga/a;
      o.x = (gb/a + rh)/ga;
    }
  }
  else if (kind == F_FC || kind == F_CHILD) {
    const int nt = (w0 >> 3) & 3;
    double pb = 0.;
#pragma unroll
    for (int t = 0; t < DIM - 1; t++)
      if (t < nt) {
	const double a = ct[(w0 >> (18 + 7*t)) & 127];
	const int cnt = (w0 >> (5 + 3*t)) & 7;
	double P;
	if (cnt == 0)
	  P = x[1 + TS*t];
	else {
	  double av = 0., n = 0.;
#pragma unroll
	  for (int k = 0; k < TS; k++)
	    if (k < cnt) {
	      n += 1.;
	      av += 1.*x[1 + TS*t + k];
	    }
	  P = av/n;
	}
	pb += a*P;
      }
    const double gc = 2.*pb/3.;
    const double c0 = ct[(w0 >> 11) & 127];
    if (kind == F_FC) {
      o.y = w*(2./3.);
      o.x = w*(c0*x[0] + gc);
    }
    else {
      o.y = w*c0;
      o.x = w*((2./3.)*x[0] - gc);
    }
  }
  else if (kind == F_SUM) {
    const int nch = (w0 >> 3) & 7;
    double na = 0., nb = 0.;
#pragma unroll
    for (int i = 0; i < (DIM == 3 ? 4 : 2); i++)
      if (i < nch) {
	na += y[i];
	nb += x[i];
      }
    if (DIM > 2) {
      na /= 4/2.;
      nb /= 4/2.;
    }
    o.x = nb; o.y = na;
  }
  else if (kind == F_GHOST)
    o.x = ct[(w0 >> 11) & 127]*x[0];
  return o;
}

// The loads of the kernel are unconditional (an idle thread loads the last record of the level and turns it
// into a NOP, an input that comes from LDS loads cell 0 from global memory and the other way round, selects
// afterwards): straight-line code in which the compiler counts the outstanding loads exactly -- behind
// branches it waits for everything, records and old values just issued included.

// the record of thread tid on level L (the plan ends with two levels of one NOP each); only the words the
// dimension uses are loaded (a loaded word that nobody reads leaves a free register with a load pending on
// it: the next write to it waits for the load)
template <int DIM>
__device__ inline FlowRec flow_load_rec (const FlowRec * rec, const int * lev_off, int L, int tid)
{
  typedef int int4v __attribute__((ext_vector_type(4)));
  typedef int int3v __attribute__((ext_vector_type(3)));
  const int a = lev_off[L], b = lev_off[L + 1];
  const bool valid = a + tid < b;
  const int * q = (const int *) (rec + (valid ? a + tid : b - 1));
  FlowRec r;
  const int4v q0 = *(const int4v *) q;
  r.w0 = valid ? (unsigned) q0.x : (unsigned) F_NOP;
  r.out_g = valid ? q0.y : -1;
  r.in[0] = valid ? q0.z : FLOW_LDS;
  r.in[1] = valid ? q0.w : FLOW_LDS;
#pragma unroll
  for (int j = 2; j < FLOW_NIN; j++) r.in[j] = FLOW_LDS;
  if (DIM == 3) {
    const int4v q1 = *(const int4v *) (q + 4);
    const int3v q2 = *(const int3v *) (q + 8);
    r.in[2] = valid ? q1.x : FLOW_LDS; r.in[3] = valid ? q1.y : FLOW_LDS;
    r.in[4] = valid ? q1.z : FLOW_LDS; r.in[5] = valid ? q1.w : FLOW_LDS;
    r.in[6] = valid ? q2.x : FLOW_LDS; r.in[7] = valid ? q2.y : FLOW_LDS; r.in[8] = valid ? q2.z : FLOW_LDS;
  }
  else {
    const int3v q1 = *(const int3v *) (q + 4);
    r.in[2] = valid ? q1.x : FLOW_LDS; r.in[3] = valid ? q1.y : FLOW_LDS; r.in[4] = valid ? q1.z : FLOW_LDS;
  }
  return r;
}

template <int DIM>
__device__ inline void flow_prefetch (const FlowRec & r, const double * u, const double * rhs, double * v)
{
#pragma unroll
  for (int j = 0; j < FlowShape<DIM>::NIN; j++)
    v[j] = u[(r.in[j] & FLOW_LDS) ? 0 : r.in[j]];
  v[FLOW_NIN] = rhs[r.out_g >= 0 ? r.out_g : 0];
}

// the whole relax loop: one workgroup, one operation per thread and level
template <int DIM>
__global__ void __launch_bounds__(FLOW_WIDTH)
t_relax_flow (const FlowRec * __restrict__ rec, const int * __restrict__ lev_off, int nlev,
	      const double * __restrict__ ctab, int nct, double * u, const double * __restrict__ rhs,
	      double omega, int op, double w)
{
  __shared__ FlowPair lo[FLOW_NBUF][FLOW_WIDTH];
  __shared__ double ct[FLOW_NCONST];
  constexpr int NIN = FlowShape<DIM>::NIN;
  const int tid = threadIdx.x;
  if (tid < nct) ct[tid] = ctab[tid];
#pragma unroll
  for (int q = 0; q < FLOW_NBUF; q++) lo[q][tid] = FlowPair { 0., 0. };
  // In level L: the record of L + 4 is loaded, the old values of L + 2 (its record came two levels ago) are
  // loaded, L is evaluated from registers and LDS.  Five records and three sets of old values rotate through
  // fixed registers: the loop is written out fifteen levels at a time, no copies (a copy of a set would
  // wait for its loads).  A load is used two levels after it was issued; vmcnt counts in issue order, so
  // nothing a level waits for is younger than two levels.  The stores of the levels before L were issued
  // before the barrier that ended L - 1: the loads of this level see them (same CU, in order).
  FlowRec r0 = flow_load_rec<DIM> (rec, lev_off, 0, tid);
  FlowRec r1 = flow_load_rec<DIM> (rec, lev_off, 1, tid);
  FlowRec r2 = flow_load_rec<DIM> (rec, lev_off, 2, tid);
  FlowRec r3 = flow_load_rec<DIM> (rec, lev_off, 3, tid);
  FlowRec r4;
  double v0[FLOW_NIN + 1], v1[FLOW_NIN + 1], v2[FLOW_NIN + 1];
  flow_prefetch<DIM> (r0, u, rhs, v0);
  flow_prefetch<DIM> (r1, u, rhs, v1);
  __syncthreads ();
#define FLOW_STEP(L, B, R0, R2, R4, V0, V2) {				\
    R4 = flow_load_rec<DIM> (rec, lev_off, (L) + 4, tid);		\
    flow_prefetch<DIM> (R2, u, rhs, V2);				\
    double x[NIN], y[NIN];						\
    _Pragma ("unroll")							\
    for (int j = 0; j < NIN; j++) {					\
      const FlowPair p = lo[0][FLOW_SLOT (R0.in[j])];			\
      x[j] = (R0.in[j] & FLOW_LDS) ? p.x : V0[j];			\
      y[j] = p.y;							\
    }									\
    const FlowPair o = flow_eval<DIM> (R0.w0, x, y, V0[FLOW_NIN], ct, omega, op, w); \
    lo[B][tid] = o;							\
    if (R0.out_g >= 0)							\
      u[R0.out_g] = o.x;						\
    __syncthreads ();							\
  }
  for (int L = 0; L < nlev; L += 15) {
    FLOW_STEP (L + 0, 0, r0, r2, r4, v0, v2);
    if (L + 1 >= nlev) break;
    FLOW_STEP (L + 1, 1, r1, r3, r0, v1, v0);
    if (L + 2 >= nlev) break;
    FLOW_STEP (L + 2, 2, r2, r4, r1, v2, v1);
    if (L + 3 >= nlev) break;
    FLOW_STEP (L + 3, 0, r3, r0, r2, v0, v2);
    if (L + 4 >= nlev) break;
    FLOW_STEP (L + 4, 1, r4, r1, r3, v1, v0);
    if (L + 5 >= nlev) break;
    FLOW_STEP (L + 5, 2, r0, r2, r4, v2, v1);
    if (L + 6 >= nlev) break;
    FLOW_STEP (L + 6, 0, r1, r3, r0, v0, v2);
    if (L + 7 >= nlev) break;
    FLOW_STEP (L + 7, 1, r2, r4, r1, v1, v0);
    if (L + 8 >= nlev) break;
    FLOW_STEP (L + 8, 2, r3, r0, r2, v2, v1);
    if (L + 9 >= nlev) break;
    FLOW_STEP (L + 9, 0, r4, r1, r3, v0, v2);
    if (L + 10 >= nlev) break;
    FLOW_STEP (L + 10, 1, r0, r2, r4, v1, v0);
    if (L + 11 >= nlev) break;
    FLOW_STEP (L + 11, 2, r1, r3, r0, v2, v1);
    if (L + 12 >= nlev) break;
    FLOW_STEP (L + 12, 0, r2, r4, r1, v0, v2);
    if (L + 13 >= nlev) break;
    FLOW_STEP (L + 13, 1, r3, r0, r2, v1, v0);
    if (L + 14 >= nlev) break;
    FLOW_STEP (L + 14, 2, r4, r1, r3, v2, v1);
  }
#undef FLOW_STEP
}

// ---- host: the plan ---------------------------------------------------------------------------

struct FlowPlan {
  int nlev = 0, nops = 0, nct = 0;
  FlowRec * rec = nullptr;         // device
  int * lev_off = nullptr;         // device, nlev + 1
  double * ct = nullptr;           // device
  std::vector<FlowRec> h_rec;      // host copies (gfship_tree_host_check)
  std::vector<int> h_lev_off;
  std::vector<double> h_ct;
};

inline void flow_free (FlowPlan & F)
{
  (void) hipFree (F.rec); (void) hipFree (F.lev_off); (void) hipFree (F.ct);
  F = FlowPlan ();
}

// the stencil of one cell read back from the streams tape_cell_gen wrote
struct FlowTerm { double a; int cnt; int idx[4]; };
struct FlowItem { double c; int g; int nt; FlowTerm t[2]; };            /* FC, or one child of a DEEP face */
struct FlowFace { int kind; int g; FlowItem fc; int nch; FlowItem ch[4]; };

inline bool flow_parse_interp (const int *& ti, const double *& td, const int *& tv, int dim, FlowItem & it)
{
  it.nt = *ti++;
  if (it.nt < 0 || it.nt > dim - 1) return false;
  for (int t = 0; t < it.nt; t++) {
    FlowTerm & T = it.t[t];
    T.a = *td++;
    T.cnt = *ti++;
    if (T.cnt < 0 || T.cnt > (dim == 3 ? 4 : 2)) return false;
    const int n = T.cnt == 0 ? 1 : T.cnt;
    for (int k = 0; k < n; k++) T.idx[k] = *tv++;
  }
  return true;
}

inline bool flow_parse_cell (const int * ti, const double * td, const int * tv, int dim, int & self, FlowFace * f)
{
  self = *tv++;
  for (int d = 0; d < 2*dim; d++) {
    FlowFace & F = f[d];
    F.kind = *ti++;
    F.nch = 0;
    if (F.kind == K_NONE) continue;
    if (F.kind == K_SAME) F.g = *tv++;
    else if (F.kind == K_FC) {
      F.fc.c = *td++;
      F.fc.g = *tv++;
      if (!flow_parse_interp (ti, td, tv, dim, F.fc)) return false;
    }
    else if (F.kind == K_DEEP) {
      F.nch = *ti++;
      if (F.nch < 0 || F.nch > (dim == 3 ? 4 : 2)) return false;
      for (int i = 0; i < F.nch; i++) {
	F.ch[i].c = *td++;
	F.ch[i].g = *tv++;
	if (!flow_parse_interp (ti, td, tv, dim, F.ch[i])) return false;
      }
    }
    else return false;
  }
  return true;
}

struct FlowBuilder {
  int dim;
  std::vector<int> wlev, wop, rlev;                 /* per cell: level and operation of its last write; last read */
  std::vector<std::vector<FlowRec>> lev;            /* lev[L - 1]: the operations of level L */
  std::vector<double> ct;
  bool ok = true;

  int constant (double c)
  {
    for (size_t i = 0; i < ct.size (); i++)
      if (!memcmp (&ct[i], &c, sizeof (double))) return (int) i;
    if (ct.size () >= FLOW_NCONST) { ok = false; return 0; }
    ct.push_back (c);
    return (int) ct.size () - 1;
  }
  static int slot (int L, int i) { return ((L - 1) % FLOW_NBUF)*FLOW_WIDTH + i; }   /* result i of level L */
  int fill (int L) const { return L >= 1 && L <= (int) lev.size () ? (int) lev[L - 1].size () : 0; }
  void need (int L) { if ((int) lev.size () < L) lev.resize (L); }
  int ready (const FlowItem & it) const           /* first level an FC / CHILD may run at */
  {
    int L = wlev[it.g];
    for (int t = 0; t < it.nt; t++)
      for (int k = 0; k < (it.t[t].cnt == 0 ? 1 : it.t[t].cnt); k++)
	L = std::max (L, wlev[it.t[t].idx[k]]);
    return L + 1;
  }
  int ref (int g, int L)                           /* the input g of an operation of level L */
  {
    rlev[g] = std::max (rlev[g], L);
    /* level 0: the values before the loop; a value at most FLOW_PD levels old is still in the LDS */
    return wlev[g] > 0 && L - wlev[g] <= FLOW_PD ? (FLOW_LDS | slot (wlev[g], wop[g])) : g;
  }
  int emit (int L, const FlowRec & r) { need (L); lev[L - 1].push_back (r); return (int) lev[L - 1].size () - 1; }
  static FlowRec blank (int kind)
  {
    FlowRec r;
    r.w0 = kind; r.out_g = -1;
    for (int j = 0; j < FLOW_NIN; j++) r.in[j] = FLOW_LDS;
    return r;
  }
  int emit_item (int kind, const FlowItem & it, int L)
  {
    const int TS = dim == 3 ? 4 : 2;
    FlowRec r = blank (kind);
    r.w0 |= it.nt << 3;
    r.w0 |= (unsigned) constant (it.c) << 11;
    r.in[0] = ref (it.g, L);
    for (int t = 0; t < it.nt; t++) {
      r.w0 |= it.t[t].cnt << (5 + 3*t);
      r.w0 |= (unsigned) constant (it.t[t].a) << (18 + 7*t);
      for (int k = 0; k < (it.t[t].cnt == 0 ? 1 : it.t[t].cnt); k++)
	r.in[1 + TS*t + k] = ref (it.t[t].idx[k], L);
    }
    return emit (L, r);
  }
  void ghost (int g, int img, double s)
  {
    int L = std::max (wlev[img], std::max (rlev[g], wlev[g])) + 1;
    while (fill (L) >= FLOW_WIDTH) L++;
    FlowRec r = blank (F_GHOST);
    r.w0 |= (unsigned) constant (s) << 11;
    r.out_g = g;
    r.in[0] = ref (img, L);
    const int i = emit (L, r);
    wlev[g] = L; wop[g] = i; rlev[g] = 0;
  }
  void cell (int g, int self, const FlowFace * f, int cell_level, bool reads_self)
  {
    int L = std::max (rlev[g], wlev[g]) + 1;
    int n1 = 0, n2 = 0;      /* operations one and two levels before the cell's */
    if (reads_self) L = std::max (L, wlev[self] + 1);
    for (int d = 0; d < 2*dim; d++) {
      const FlowFace & F = f[d];
      if (F.kind == K_SAME) L = std::max (L, wlev[F.g] + 1);
      else if (F.kind == K_FC) { L = std::max (L, ready (F.fc) + 1); n1++; }
      else if (F.kind == K_DEEP) {
	for (int i = 0; i < F.nch; i++) { L = std::max (L, ready (F.ch[i]) + 2); n2++; }
	n1++;
      }
    }
    if (n2) L = std::max (L, 3); else if (n1) L = std::max (L, 2);
    while (fill (L) + 1 > FLOW_WIDTH || (n1 && fill (L - 1) + n1 > FLOW_WIDTH) || (n2 && fill (L - 2) + n2 > FLOW_WIDTH))
      L++;
    FlowRec r = blank (F_CELL);
    r.w0 |= (unsigned) cell_level << 15;
    r.out_g = g;
    if (reads_self) { r.w0 |= 1u << 20; r.in[0] = ref (self, L); }
    for (int d = 0; d < 2*dim; d++) {
      const FlowFace & F = f[d];
      if (F.kind == K_SAME) {
	r.w0 |= FK_SAME << (3 + 2*d);
	r.in[1 + d] = ref (F.g, L);
      }
      else if (F.kind == K_FC) {
	r.w0 |= FK_PAIR << (3 + 2*d);
	r.in[1 + d] = FLOW_LDS | slot (L - 1, emit_item (F_FC, F.fc, L - 1));
      }
      else if (F.kind == K_DEEP) {
	FlowRec s = blank (F_SUM);
	s.w0 |= F.nch << 3;
	for (int i = 0; i < F.nch; i++)
	  s.in[i] = FLOW_LDS | slot (L - 2, emit_item (F_CHILD, F.ch[i], L - 2));
	r.w0 |= FK_PAIR << (3 + 2*d);
	r.in[1 + d] = FLOW_LDS | slot (L - 1, emit (L - 1, s));
      }
    }
    const int i = emit (L, r);
    wlev[g] = L; wop[g] = i; rlev[g] = 0;
  }
};

// the loop of nrelax sweeps of Sweep S (its cells in the order S->h_g, a valid sequential order) with
// the copies of the ghosts between the sweeps, as a flow plan; returns false when the loop does not fit
// the format (the caller keeps the tape kernels)
inline bool flow_plan (int ncell, int dim, const Sweep * S, unsigned nrelax, const Sgn6 & sg, const int * cell_level_of,
		       bool reads_self, FlowPlan * out, bool host_only)
{
  flow_free (*out);
  FlowBuilder B;
  B.dim = dim;
  B.wlev.assign (ncell, 0); B.wop.assign (ncell, 0); B.rlev.assign (ncell, 0);
  const size_t nc = S->h_g.size ();
  std::vector<FlowFace> faces (nc*6);
  std::vector<int> selfs (nc);
  for (size_t c = 0; c < nc; c++)
    if (!flow_parse_cell (S->h_ti.data () + S->h_cell_off[3*c], S->h_td.data () + S->h_cell_off[3*c + 1],
			  S->h_tv.data () + S->h_cell_off[3*c + 2], dim, selfs[c], &faces[6*c]))
      return false;
  for (unsigned sw = 0; sw < nrelax && B.ok; sw++) {
    for (const Ghost & G : S->h_ghosts) B.ghost (G.g, G.img, sg.s[G.side]);
    for (size_t c = 0; c < nc && B.ok; c++)
      B.cell (S->h_g[c], selfs[c], &faces[6*c], cell_level_of[S->h_g[c]], reads_self);
  }
  if (!B.ok) return false;
  FlowPlan & F = *out;
  F.nlev = (int) B.lev.size ();
  std::vector<FlowRec> rec;
  std::vector<int> off (1, 0);
  for (const auto & l : B.lev) {
    rec.insert (rec.end (), l.begin (), l.end ());
    off.push_back ((int) rec.size ());
  }
  F.nops = (int) rec.size ();
  for (int k = 0; k < FLOW_PD + 2; k++) {        /* the records the kernel loads ahead of the last level */
    rec.push_back (FlowBuilder::blank (F_NOP));
    off.push_back ((int) rec.size ());
  }
  F.nct = (int) B.ct.size ();
  if (host_only) { F.h_rec = rec; F.h_lev_off = off; F.h_ct = B.ct; return true; }
  if (rec.empty ()) return false;
  std::vector<double> ct = B.ct;
  if (ct.empty ()) ct.push_back (0.);
  if (hipMalloc ((void **) &F.rec, rec.size ()*sizeof (FlowRec)) != hipSuccess ||
      hipMalloc ((void **) &F.lev_off, off.size ()*sizeof (int)) != hipSuccess ||
      hipMalloc ((void **) &F.ct, ct.size ()*sizeof (double)) != hipSuccess ||
      hipMemcpy (F.rec, rec.data (), rec.size ()*sizeof (FlowRec), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy (F.lev_off, off.data (), off.size ()*sizeof (int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy (F.ct, ct.data (), ct.size ()*sizeof (double), hipMemcpyHostToDevice) != hipSuccess) {
    flow_free (F);
    return false;
  }
  return true;
}

// the kernel's program on the host, with its timing: the global inputs of level L are read BEFORE the
// stores of level L - 1 (they are loaded a level ahead), the operations of a level run backwards, the
// stores of a level land after its loads.  Returns the number of hazards: a store of level L - 1 or L
// onto a cell that level L reads from global memory (which the schedule must exclude).
inline long long flow_emulate (const FlowPlan & F, int dim, std::vector<double> & u, const std::vector<double> & rhs,
			       double omega, int op, double w)
{
  long long hazards = 0;
  std::vector<FlowPair> lo ((size_t) FLOW_NBUF*FLOW_WIDTH, FlowPair { 0., 0. });
  std::vector<std::vector<double>> pre (FLOW_NBUF, std::vector<double> ((size_t) FLOW_WIDTH*(FLOW_NIN + 1)));
  std::vector<int> stored_at (u.size (), -10);
  auto prefetch = [&] (int L, std::vector<double> & v) {      /* what flow_prefetch loads for level L (0-based) */
    if (L >= F.nlev) return;
    for (int i = F.h_lev_off[L]; i < F.h_lev_off[L + 1]; i++) {
      const FlowRec & r = F.h_rec[i];
      double * vi = &v[(size_t) (i - F.h_lev_off[L])*(FLOW_NIN + 1)];
      for (int j = 0; j < FLOW_NIN; j++)
	vi[j] = (r.in[j] & FLOW_LDS) ? 0. : u[r.in[j]];
      vi[FLOW_NIN] = ((r.w0 & 7) == F_CELL) ? rhs[r.out_g] : 0.;
    }
  };
  for (int L = 0; L < FLOW_PD; L++) prefetch (L, pre[L % FLOW_NBUF]);
  for (int L = 0; L < F.nlev; L++) {
    prefetch (L + FLOW_PD, pre[(L + FLOW_PD) % FLOW_NBUF]);
    std::vector<std::pair<int, double>> stores;
    std::vector<FlowPair> cur (FLOW_WIDTH, FlowPair { 0., 0. });
    for (int i = F.h_lev_off[L + 1]; i-- > F.h_lev_off[L]; ) {
      const FlowRec & r = F.h_rec[i];
      const double * vi = &pre[L % FLOW_NBUF][(size_t) (i - F.h_lev_off[L])*(FLOW_NIN + 1)];
      double x[FLOW_NIN], y[FLOW_NIN];
      for (int j = 0; j < FLOW_NIN; j++) {
	const int sl = FLOW_SLOT (r.in[j]);
	if ((r.in[j] & FLOW_LDS) && sl/FLOW_WIDTH == L % FLOW_NBUF && (r.w0 & 7) != F_NOP && sl != 0)
	  hazards++;       /* an input in the buffer this level writes */
	const FlowPair p = lo[sl];
	x[j] = (r.in[j] & FLOW_LDS) ? p.x : vi[j];
	y[j] = p.y;
      }
      const FlowPair o = dim == 3 ? flow_eval<3> (r.w0, x, y, vi[FLOW_NIN], F.h_ct.data (), omega, op, w) :
	flow_eval<2> (r.w0, x, y, vi[FLOW_NIN], F.h_ct.data (), omega, op, w);
      cur[i - F.h_lev_off[L]] = o;
      if (r.out_g >= 0) stores.push_back ({ r.out_g, o.x });
      for (int j = 0; j < FLOW_NIN; j++)
	if (!(r.in[j] & FLOW_LDS) && stored_at[r.in[j]] >= L - FLOW_PD)
	  hazards++;         /* the load was issued in level L - FLOW_PD: a store since then may or may not be seen */
    }
    for (auto & s : stores) {
      if (stored_at[s.first] == L) hazards++;      /* two stores of one level onto one cell */
      u[s.first] = s.second;
      stored_at[s.first] = L;
    }
    // a store of this level onto a cell this level reads from global memory (the schedule keeps a write
    // one level behind the reads of the value it replaces)
    for (int i = F.h_lev_off[L]; i < F.h_lev_off[L + 1]; i++)
      for (int j = 0; j < FLOW_NIN; j++) {
	const int g = F.h_rec[i].in[j];
	if (!(g & FLOW_LDS) && stored_at[g] == L && g != F.h_rec[i].out_g) hazards++;
      }
    for (int i = 0; i < FLOW_WIDTH; i++) lo[(size_t) (L % FLOW_NBUF)*FLOW_WIDTH + i] = cur[i];
  }
  return hazards;
}
