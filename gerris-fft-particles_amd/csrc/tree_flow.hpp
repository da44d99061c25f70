// tree_flow.hpp -- the relax loop of a tree level (src/poisson.c:1070-1089 over the cells of
// T_LEVEL_LEAFS, src/poisson.c:604-632) as a dataflow program of fixed-format micro-operations.
// Included by tree.hip inside its anonymous namespace (after Sweep, Ghost, Sgn6, the K_* codes).
//
// Why: t_relax_nodes interprets the stencil of a cell from a variable-length tape, one thread per
// cell -- a coarse leaf with fine neighbours is a chain of ~100 dependent LDS reads, and the slowest
// thread of a dependency level sets its time (8-10 us per level on the fine levels of the octree
// bench).  Here the stencil of a cell is cut where its expression tree
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
// before their CELL.  A level holds at most `width' (<= 512) operations: thread t of the workgroup executes
// operation t of the level and leaves its result in LDS (a ring of FLOW_NBUF buffers indexed by the level)
// for the operations of the next levels -- that is how the pairs travel, and how a cell value reaches
// the neighbours that follow it closely.  Older values are read from global memory one level ahead of
// their use (the workgroup's own stores of earlier levels are ordered before these loads by the barrier:
// same CU, no acknowledgement needed); the record of a level is loaded three levels ahead.  An extra
// wavefront does the stores, a level late, from the LDS.  What is left on the chain of a level: the LDS
// reads of its fresh inputs, the arithmetic of ONE micro-operation, an LDS write, a barrier.
//
// The loop runs on copies of the values and of the right-hand side laid out in the order of the plan
// (t_flow_pack / t_flow_unpack around it): see FlowPlan.
//
// The schedule keeps the order of the sequential program as loop_plan does (a read follows the write
// it must see, a write follows the reads of the value it replaces, one level at least), so that the
// single image in global memory is right for every load; gfship_tree_host_check runs the plan on the
// host with the kernel's timing of loads and stores (and its choice of the branch-free arithmetic per
// wavefront) and compares with the reference's program.
//
// Measured (octree bench, tools/tree_bench.py 3 4 2: 39 712 leaves, levels 4-6; profiles/r03_tree_flow.txt;
// DESIGN.md 11.15 has the whole account): the loop of the finest level 3.43 ms (t_relax_nodes_pf, 349 levels of
// the tape plan) -> 0.79 ms (681 levels of 512 operations), the cycle 6.6 -> 1.5 ms.  A level costs 1.0-1.2 us
// whatever its width: every wavefront of a level wants the vector memory pipeline (14 loads per thread), then the
// LDS, then the vector ALU (~100 instructions around ~40 f64 operations and one division) and the barrier keeps
// them in step.  What paid, in order: fixed-format operations with independent loads instead of the tape (2.7 x);
// even and odd wavefronts in different orders (17 %); the kinds of a level at multiples of 64, i.e. no wavefront
// with two kinds (14 %); the working copies in plan order (18 % on the finest loop).  Tried without a gain, and left
// in or out as said: loads two levels ahead (out: 15-fold unrolling, 224 VGPRs); the storing wavefront (in); the
// working copies read into the L2 once at the start (in); the lines of the record stream touched eight levels
// ahead by LDS-DMA loads (out); 128 / 256 / 384 operations per level (GFSHIP_FLOW_WIDTH; 256 is the default on
// quadtrees); on quadtrees a CELL with the FC / CHILD / SUM of its faces in the same operation (out: 38 % fewer
// levels, each twice as expensive); a select-free CELL / FC (out: 5 % slower).

enum { F_CELL = 0, F_FC = 1, F_CHILD = 2, F_SUM = 3, F_GHOST = 4, F_NOP = 5 };
enum { FK_NONE = 0, FK_SAME = 1, FK_PAIR = 2 };
#define FLOW_WIDTH  512             /* most operations per level = threads of the workgroup; a plan may use fewer */
#define FLOW_PD     1               /* old values are loaded FLOW_PD levels ahead, records FLOW_PD + 2 */
#define FLOW_NBUF   (FLOW_PD + 2)   /* a result stays in the LDS for FLOW_PD + 1 levels: its store is issued a level late */
#define FLOW_NIN    10
#define FLOW_NCONST 128
// an input of a record: < 0: the result of an operation of one of the last FLOW_PD levels, bits 4-14 = its
// slot (level % FLOW_NBUF)*FLOW_WIDTH + operation (i.e. bits 0-14 = its byte address in the LDS);
// >= 0: a cell, as the byte offset 8 g of its value in global memory.  out_g: the same offset, or -1
#define FLOW_IS_LDS(ref)  ((ref) < 0)
#define FLOW_LDS_REF(slot) ((int) (0x80000000u | ((unsigned) (slot) << 4)))
#define FLOW_LDS_ADDR(ref) ((ref) & 0x7ff0)
#define FLOW_NONE   FLOW_LDS_REF (FLOW_LDS_SLOTS - 1)      /* no input: a slot no operation writes */
#define FLOW_LDS_SLOTS 2048         /* 32 KB: any masked address stays inside */
#define FLOW_MAXLEV  5900            /* levels of a plan (their first operations sit in the LDS) */

struct __attribute__((aligned(16))) FlowRec { unsigned w0; int out_g; int in[FLOW_NIN]; };
// w0: bits 0-2 the kind
//   CELL   3-14 the kind of each face (FK_*), 15-19 the level of the cell, 20: reads its own value
//          in[0] the cell itself, in[1 + d] the neighbour (FK_SAME) or the pair of the face (FK_PAIR)
//   FC / CHILD  3-4 terms of the interpolation, 5-7 / 8-10 cells averaged in term 0 / 1 (0: one
//          value), 11-17 / 18-24 / 25-31 the constants cb (gbi), a0, a1 in the table
//          in[0] the coarse neighbour (the child), in[1 + TS t + k] value k of term t
//   SUM    3-5 children;  in[i] the pair of child i
//   GHOST  11-17 the constant (the sign of the homogeneous condition);  in[0] the image
// The operations of a level are sorted by kind (CELL | FC, CHILD | SUM, GHOST), each kind starting at a multiple of
// 64 when the plan is 256 wide at least (the schedule keeps the room): a wavefront runs one branch of flow_eval

template <int DIM> struct FlowShape {
  static constexpr int TS = DIM == 3 ? 4 : 2;            /* values of one interpolation term */
  static constexpr int NIN = DIM == 3 ? 9 : 5;           /* inputs a record of this dimension uses */
};

struct __attribute__((aligned(16))) FlowPair { double x, y; };

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
	const double t = gb - rh;
	const double q = ga == 4. ? t*0.25 : t/ga;      /* t/4 and t*0.25 are the same number */
	if (DIM == 2)
	  r = (1. - omega)*x[0] + omega*q;
	else
	  r = q;
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
	  /* av/n; n = 1, 2, 4: the product by 1/n is the same number, without the division */
	  P = n == 4. ? av*0.25 : n == 2. ? av*0.5 : n == 1. ? av : av/n;
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

// The same operations without a branch, for a wavefront whose lanes all run the same kind (the operations of
// a level are sorted by kind): what a lane does not have is added as + 0. -- the sums start from + 0. and a
// sum that starts from + 0. is never - 0., so x + 0. = x bit for bit -- and selected afterwards.  The branches
// of flow_eval cost more than its arithmetic (46 exec-mask branches per operation).  Divisions: one per
// operation (t/ga; 2 pb/3); av/n is a product for n = 1, 2, 4 and a division, behind a branch the whole
// wavefront takes together, when a lane has n = 3.
template <int DIM, int KIND, class ANY>
__host__ __device__ inline FlowPair flow_eval_uniform (unsigned w0, const double * x, const double * y, double rh,
						       const double * ct, double omega, int op, double w, const ANY & any)
{
  constexpr int TS = FlowShape<DIM>::TS;
  FlowPair o = { 0., 0. };
  if (KIND == F_CELL) {
    double ga = 0., gb = 0.;
#pragma unroll
    for (int d = 0; d < 2*DIM; d++) {
      const int fk = (w0 >> (3 + 2*d)) & 3;
      const double na = fk == FK_SAME ? w : fk == FK_PAIR ? y[1 + d] : 0.;
      const double nb = fk == FK_SAME ? w*x[1 + d] : fk == FK_PAIR ? x[1 + d] : 0.;
      ga += na; gb += nb;
    }
    if (op == 0) {
      const double q = (gb - rh)/ga;
      const double r = DIM == 2 ? (1. - omega)*x[0] + omega*q : q;
      o.x = ga != 0. ? r : 0.;
    }
    else {
      const int l = (w0 >> 15) & 31;
      const double h = 1./(1 << l);
      const double a = 1.*h*h;
      ga = 1. + ga/a;
      o.x = (gb/a + rh)/ga;
    }
  }
  else if (KIND == F_FC) {      /* FC and CHILD */
    const int nt = (w0 >> 3) & 3;
    double pb = 0.;
#pragma unroll
    for (int t = 0; t < DIM - 1; t++) {
      const double a = ct[(w0 >> (18 + 7*t)) & 127];
      const int cnt = (w0 >> (5 + 3*t)) & 7;
      double av = 0.;
#pragma unroll
      for (int k = 0; k < TS; k++)
	av += k < cnt ? 1.*x[1 + TS*t + k] : 0.;
      double P = cnt == 4 ? av*0.25 : cnt == 2 ? av*0.5 : av;      /* av/n, n = 4, 2, 1 */
      if (any (t < nt && cnt == 3)) {
	const double n3 = 3.;
	P = cnt == 3 ? av/n3 : P;
      }
      P = cnt == 0 ? x[1 + TS*t] : P;
      pb += t < nt ? a*P : 0.;
    }
    const double gc = 2.*pb/3.;
    const double c0 = ct[(w0 >> 11) & 127];
    const bool fc = (w0 & 7) == F_FC;
    o.y = fc ? w*(2./3.) : w*c0;
    o.x = fc ? w*(c0*x[0] + gc) : w*((2./3.)*x[0] - gc);
  }
  else if (KIND == F_SUM) {
    const int nch = (w0 >> 3) & 7;
    double na = 0., nb = 0.;
#pragma unroll
    for (int i = 0; i < (DIM == 3 ? 4 : 2); i++) {
      na += i < nch ? y[i] : 0.;
      nb += i < nch ? x[i] : 0.;
    }
    if (DIM > 2) {
      na /= 4/2.;
      nb /= 4/2.;
    }
    o.x = nb; o.y = na;
  }
  else if (KIND == F_GHOST)
    o.x = ct[(w0 >> 11) & 127]*x[0];
  return o;
}

// the kind all the operations of a wavefront share (NOPs apart), or -1
__host__ __device__ inline int flow_class (int kind) { return kind == F_CHILD ? F_FC : kind; }

struct FlowAnyDev {
  __device__ inline bool operator() (bool b) const { return __any (b); }
};

// One level of one wavefront: the inputs from the LDS, the arithmetic.  A wavefront of one kind reads only
// what that kind can use (FC / CHILD: nine values; CELL: the values and the pairs of its faces; SUM: pairs) --
// the reads of the LDS are a third of the time of a level when every lane reads ten pairs.
template <int DIM>
__device__ inline FlowPair flow_eval_wave (const FlowRec & r, const double * v, const FlowPair * lo,
					   const double * ct, double omega, int op, double w)
{
  constexpr int NIN = FlowShape<DIM>::NIN;
  const unsigned w0 = r.w0;
  const int kind = flow_class (w0 & 7);
  const unsigned long long busy = __ballot (kind != F_NOP);
  FlowPair o = { 0., 0. };
  if (busy == 0ull)
    return o;
  const int first = __builtin_amdgcn_readlane (kind, __ffsll ((long long) busy) - 1);
  const FlowAnyDev any;
  double x[NIN], y[NIN];
#define FLOW_X(j) { const double px = *(const double *) ((const char *) lo + FLOW_LDS_ADDR (r.in[j])); \
    x[j] = FLOW_IS_LDS (r.in[j]) ? px : v[j]; y[j] = 0.; }
#define FLOW_XY(j) { const FlowPair p = *(const FlowPair *) ((const char *) lo + FLOW_LDS_ADDR (r.in[j])); \
    x[j] = FLOW_IS_LDS (r.in[j]) ? p.x : v[j]; y[j] = p.y; }
  if (__ballot (kind != F_NOP && kind != first) != 0ull) {
#pragma unroll
    for (int j = 0; j < NIN; j++) FLOW_XY (j)
    return flow_eval<DIM> (w0, x, y, v[FLOW_NIN], ct, omega, op, w);
  }
#pragma unroll
  for (int j = 0; j < NIN; j++) { x[j] = 0.; y[j] = 0.; }
  if (first == F_CELL) {
    if (DIM == 2) FLOW_X (0)
#pragma unroll
    for (int j = 1; j <= 2*DIM; j++) FLOW_XY (j)
    return flow_eval_uniform<DIM, F_CELL> (w0, x, y, v[FLOW_NIN], ct, omega, op, w, any);
  }
  if (first == F_FC) {
#pragma unroll
    for (int j = 0; j < 1 + (DIM - 1)*FlowShape<DIM>::TS; j++) FLOW_X (j)
    return flow_eval_uniform<DIM, F_FC> (w0, x, y, v[FLOW_NIN], ct, omega, op, w, any);
  }
  if (first == F_SUM) {
#pragma unroll
    for (int j = 0; j < (DIM == 3 ? 4 : 2); j++) FLOW_XY (j)
    return flow_eval_uniform<DIM, F_SUM> (w0, x, y, v[FLOW_NIN], ct, omega, op, w, any);
  }
  FLOW_X (0)
  return flow_eval_uniform<DIM, F_GHOST> (w0, x, y, v[FLOW_NIN], ct, omega, op, w, any);
#undef FLOW_X
#undef FLOW_XY
}

// The loads of the kernel are unconditional (an idle thread loads the last record of the level and turns it
// into a NOP, an input that comes from LDS loads cell 0 from global memory and the other way round, selects
// afterwards): straight-line code in which the compiler counts the outstanding loads exactly -- behind
// branches it waits for everything, records and old values just issued included.

// the record of thread tid on level L (the plan ends with NOP levels for the loads ahead of the last level);
// only the words the dimension uses are loaded (a loaded word that nobody reads leaves a free register with a
// load pending on it: the next write to it waits for the load)
template <int DIM>
__device__ inline FlowRec flow_load_rec (const FlowRec * rec, const int * loff, int L, int tid)
{
  typedef int int4v __attribute__((ext_vector_type(4)));
  typedef int int3v __attribute__((ext_vector_type(3)));
  const int a = loff[L], b = loff[L + 1];      /* from the LDS: from global memory the load of the record would wait for this one */
  const bool valid = a + tid < b;
  const int * q = (const int *) (rec + a + tid);      /* an idle lane reads on (the array ends with spare records): every
							 lane at the same address is the slowest access there is */
  FlowRec r;
  const int4v q0 = *(const int4v *) q;
  r.w0 = valid ? (unsigned) q0.x : (unsigned) F_NOP;
  r.out_g = valid ? q0.y : -1;
  r.in[0] = valid ? q0.z : FLOW_NONE;
  r.in[1] = valid ? q0.w : FLOW_NONE;
#pragma unroll
  for (int j = 2; j < FLOW_NIN; j++) r.in[j] = FLOW_NONE;
  if (DIM == 3) {
    const int4v q1 = *(const int4v *) (q + 4);
    const int3v q2 = *(const int3v *) (q + 8);
    r.in[2] = valid ? q1.x : FLOW_NONE; r.in[3] = valid ? q1.y : FLOW_NONE;
    r.in[4] = valid ? q1.z : FLOW_NONE; r.in[5] = valid ? q1.w : FLOW_NONE;
    r.in[6] = valid ? q2.x : FLOW_NONE; r.in[7] = valid ? q2.y : FLOW_NONE; r.in[8] = valid ? q2.z : FLOW_NONE;
  }
  else {
    const int3v q1 = *(const int3v *) (q + 4);
    r.in[2] = valid ? q1.x : FLOW_NONE; r.in[3] = valid ? q1.y : FLOW_NONE; r.in[4] = valid ? q1.z : FLOW_NONE;
  }
  return r;
}

template <int DIM>
__device__ inline void flow_prefetch (const FlowRec & r, const double * u, const double * rhs, double * v, int idle)
{
  /* idle: 8 tid -- an input that is not in global memory loads a place of its own (the copies have 576 at least) */
#pragma unroll
  for (int j = 0; j < FlowShape<DIM>::NIN; j++)
    v[j] = *(const double *) ((const char *) u + (unsigned) (r.in[j] < 0 ? idle : r.in[j]));
  v[FLOW_NIN] = *(const double *) ((const char *) rhs + (unsigned) (r.out_g < 0 ? idle : r.out_g));
}

// the whole relax loop: one workgroup, one operation per thread and level
template <int DIM>
__global__ void __launch_bounds__(FLOW_WIDTH + 64)
t_relax_flow (const FlowRec * __restrict__ rec, const int * __restrict__ lev_off, int nlev,
	      const double * __restrict__ ctab, int nct, double * u, const double * __restrict__ rhs, int npos,
	      double omega, int op, double w)
{
  __shared__ FlowPair lo[FLOW_LDS_SLOTS];
  __shared__ int lout[FLOW_NBUF*FLOW_WIDTH];
  __shared__ int loff[FLOW_MAXLEV + FLOW_PD + 4];
  __shared__ double ct[FLOW_NCONST];
  const int tid = threadIdx.x;
  if (tid < nct) ct[tid] = ctab[tid];
  const int width = blockDim.x - 64;      /* the width the plan was made for; the last wavefront stores */
  for (int q = tid; q < FLOW_LDS_SLOTS; q += blockDim.x) lo[q] = FlowPair { 0., 0. };
  for (int q = tid; q < FLOW_NBUF*FLOW_WIDTH; q += blockDim.x) lout[q] = -1;
  for (int q = tid; q < nlev + FLOW_PD + 3; q += blockDim.x) loff[q] = lev_off[q];
  // The copies the loop works on were written by other compute units: the first load of each of their lines
  // comes from memory (a microsecond), and a level cannot be shorter than the loads it issued a level ago.
  // Read them once, all lanes, pipelined: afterwards they are in this XCD's L2 (and the stores of the loop
  // update them there).
  double warm = 0.;
  for (int q = tid*16; q < npos; q += blockDim.x*16)
    warm += ((const volatile double *) u)[q] + ((const volatile double *) rhs)[q];
  if (warm == 1.2345678e-300) ct[FLOW_NCONST - 1] = warm;      /* (keeps the loads) */
  __syncthreads ();
  if (tid >= width) {
    // The storing wavefront: the wavefronts that compute leave result and place in the LDS, and this one
    // stores the results of level L - 1 while level L is computed.  vmcnt counts loads and stores together,
    // in issue order: a wavefront that stored its own results would not get at the values it loaded
    // afterwards before those stores are acknowledged.
    const int lane = tid - width;
    __syncthreads ();
    for (int L = 0; L <= nlev; L++) {
      if (L >= 1) {
	const int base = ((L - 1) % FLOW_NBUF)*width;
	int g[FLOW_WIDTH/64];
	double xv[FLOW_WIDTH/64];
#pragma unroll
	for (int i = 0; i < FLOW_WIDTH/64; i++) {      /* every read of the LDS first, then the stores */
	  const int k = lane + 64*i;
	  g[i] = lout[base + (k < width ? k : 0)];
	  xv[i] = lo[base + (k < width ? k : 0)].x;
	  if (k >= width) g[i] = -1;
	}
#pragma unroll
	for (int i = 0; i < FLOW_WIDTH/64; i++)
	  if (g[i] >= 0)
	    *(double *) ((char *) u + (unsigned) g[i]) = xv[i];
      }
      if (L < nlev) __syncthreads ();
    }
    return;
  }
  // In level L: the record of L + 3 is loaded, the old values of L + 1 (its record came two levels ago) are
  // loaded, L is evaluated from registers and LDS.  Four records and two sets of old values rotate through
  // fixed registers: the loop is written out four levels at a time, no copies (a copy of a set would wait
  // for its loads).  vmcnt counts in issue order: what a level waits for was issued a level ago at least.
  // The stores of the levels before L - 1 were issued before the barrier that ended L - 1: the loads of this
  // level see them (same CU, in order); younger values come from the LDS.
  FlowRec r0 = flow_load_rec<DIM> (rec, loff, 0, tid);
  FlowRec r1 = flow_load_rec<DIM> (rec, loff, 1, tid);
  FlowRec r2 = flow_load_rec<DIM> (rec, loff, 2, tid);
  FlowRec r3;
  double v0[FLOW_NIN + 1], v1[FLOW_NIN + 1];
  flow_prefetch<DIM> (r0, u, rhs, v0, 8*tid);
  flow_prefetch<DIM> (r1, u, rhs, v1, 8*tid);      /* (loaded again in level 0; keeps v1 defined for the lab switches) */
  __syncthreads ();
  // FLOW_LAB (lab only): 1 no arithmetic, 2 no loads of old values, 4 no loads of records (wrong results);
  // 32 the times of the parts of eight levels printed (s_memtime); 64 every wavefront in the same order
#ifndef FLOW_LAB
#define FLOW_LAB 0
#endif
#if FLOW_LAB & 32
#define FLOW_TS(L, k) if ((tid & 63) == 0 && (L) >= 300 && (L) < 308)	\
    tstamp[tid >> 6][(L) - 300][k] = __builtin_amdgcn_s_memtime ();
#else
#define FLOW_TS(L, k)
#endif
#define FLOW_LOADS(L, R0, R1, R3, V1)					\
    if (!(FLOW_LAB & 2)) flow_prefetch<DIM> (R1, u, rhs, V1, 8*tid);	\
    if (!(FLOW_LAB & 4)) R3 = flow_load_rec<DIM> (rec, loff, (L) + 3, tid); else R3 = R0;
#define FLOW_COMPUTE(L, R0, V0) {					\
    FLOW_TS (L, 2)							\
    FlowPair o = { V0[0] + V0[1], V0[2] };				\
    if (!(FLOW_LAB & 1)) o = flow_eval_wave<DIM> (R0, V0, lo, ct, omega, op, w); \
    if ((FLOW_LAB & 32) && o.x == 1.2345e-300) ct[FLOW_NCONST - 4] = o.x; \
    FLOW_TS (L, 3)							\
    lo[buf*width + tid] = o;						\
    lout[buf*width + tid] = R0.out_g;					\
    buf = buf == FLOW_NBUF - 1 ? 0 : buf + 1;				\
  }
  // Two orders of a level.  Every wavefront of a level wants the vector memory pipeline, then the LDS, then
  // the vector ALU, and the barrier keeps them in step: three units, one busy at a time (measured with
  // s_memtime, FLOW_LAB=32: 900 cycles to issue the 14 loads of a level, 850 for the reads of the LDS, 950 to
  // 1900 of arithmetic, the rest waiting for the slowest).  So the odd wavefronts issue their loads AFTER
  // their arithmetic (their old values then have the barrier and the next level's LDS reads to arrive):
  // they compute while the even ones load, and the other way round.
#define FLOW_STEP_A(L, R0, R1, R3, V0, V1) {				\
    FLOW_TS (L, 0)							\
    FLOW_LOADS (L, R0, R1, R3, V1)					\
    FLOW_TS (L, 1)							\
    FLOW_COMPUTE (L, R0, V0)						\
    __syncthreads ();							\
    FLOW_TS (L, 4)							\
  }
#define FLOW_STEP_B(L, R0, R1, R3, V0, V1) {				\
    FLOW_TS (L, 0)							\
    FLOW_TS (L, 1)							\
    FLOW_COMPUTE (L, R0, V0)						\
    FLOW_LOADS (L, R0, R1, R3, V1)					\
    __syncthreads ();							\
    FLOW_TS (L, 4)							\
  }
  int buf = 0;
#if FLOW_LAB & 32
  __shared__ unsigned long long tstamp[8][8][5];
#endif
  if ((tid >> 6) & 1 & !(FLOW_LAB & 64)) {
    for (int L = 0; L < nlev; L += 4) {
      FLOW_STEP_B (L, r0, r1, r3, v0, v1);
      if (L + 1 >= nlev) break;
      FLOW_STEP_B (L + 1, r1, r2, r0, v1, v0);
      if (L + 2 >= nlev) break;
      FLOW_STEP_B (L + 2, r2, r3, r1, v0, v1);
      if (L + 3 >= nlev) break;
      FLOW_STEP_B (L + 3, r3, r0, r2, v1, v0);
    }
  }
  else {
    for (int L = 0; L < nlev; L += 4) {
      FLOW_STEP_A (L, r0, r1, r3, v0, v1);
      if (L + 1 >= nlev) break;
      FLOW_STEP_A (L + 1, r1, r2, r0, v1, v0);
      if (L + 2 >= nlev) break;
      FLOW_STEP_A (L + 2, r2, r3, r1, v0, v1);
      if (L + 3 >= nlev) break;
      FLOW_STEP_A (L + 3, r3, r0, r2, v1, v0);
    }
  }
#undef FLOW_STEP_A
#undef FLOW_STEP_B
#undef FLOW_LOADS
#undef FLOW_COMPUTE
#if FLOW_LAB & 32
  __syncthreads ();
  if (tid == 0 && nlev > 320)
    for (int wv = 0; wv < width/64; wv++)
      for (int q = 0; q < 8; q++) {
	const unsigned long long * t = tstamp[wv][q];
	printf ("flow lab: wave %d level %d kind %d: issue %llu gather %llu eval %llu barrier %llu | next start +%llu\n", wv, 300 + q,
		(int) (rec[loff[300 + q] + min (wv*64, loff[301 + q] - loff[300 + q] - 1)].w0 & 7),
		t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], q < 7 ? tstamp[wv][q + 1][0] - t[4] : 0ull);
      }
#endif
}

// into / out of the order of the plan
__global__ void t_flow_pack (const int * __restrict__ gidx, int n, const double * __restrict__ u,
			     const double * __restrict__ rhs, double * __restrict__ up, double * __restrict__ rp)
{
  const int p = blockIdx.x*blockDim.x + threadIdx.x;
  if (p < n) {
    const int g = gidx[p];
    up[p] = u[g];
    rp[p] = rhs[g];
  }
}

__global__ void t_flow_unpack (const int * __restrict__ gidx, int n, const double * __restrict__ up, double * __restrict__ u)
{
  const int p = blockIdx.x*blockDim.x + threadIdx.x;
  if (p < n)
    u[gidx[p]] = up[p];
}

// ---- host: the plan ---------------------------------------------------------------------------

struct FlowPlan {
  int nlev = 0, nops = 0, nct = 0;
  int width = FLOW_WIDTH;          // operations per level at most = threads of the launch
  FlowRec * rec = nullptr;         // device
  int * lev_off = nullptr;         // device, nlev + 1
  double * ct = nullptr;           // device
  // the loop works on copies of the values and of the right-hand side in the ORDER OF THE PLAN (the cells in
  // the order their operations first appear): the cells of a level, and their neighbours on the next, are
  // then next to each other in memory -- in the tree's own arrays (one dense array per level) the cells of a
  // dependency level lie on a skew hyperplane and every 8-byte value costs a 128-byte line from the L2
  int npos = 0;
  int * gidx = nullptr;            // device, npos: the cell at each place
  double * up = nullptr, * rp = nullptr;   // device, npos
  std::vector<int> h_gidx;
  std::vector<FlowRec> h_rec;      // host copies (gfship_tree_host_check)
  std::vector<int> h_lev_off;
  std::vector<double> h_ct;
};

inline void flow_free (FlowPlan & F)
{
  (void) hipFree (F.rec); (void) hipFree (F.lev_off); (void) hipFree (F.ct);
  (void) hipFree (F.gidx); (void) hipFree (F.up); (void) hipFree (F.rp);
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
  int width = FLOW_WIDTH;
  std::vector<int> wlev, wop, rlev;                 /* per cell: level and operation of its last write; last read */
  std::vector<std::vector<FlowRec>> lev;            /* lev[L - 1]: the operations of level L */
  std::vector<std::array<int, 3>> cls;              /* of which CELL | FC, CHILD | SUM, GHOST */
  bool padded = true;                               /* a level has room for every kind to start at a multiple of 64 */
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
  int slot (int L, int i) const { return ((L - 1) % FLOW_NBUF)*width + i; }   /* result i of level L */
  int fill (int L) const { return L >= 1 && L <= (int) lev.size () ? (int) lev[L - 1].size () : 0; }
  void need (int L) { if ((int) lev.size () < L) { lev.resize (L); cls.resize (L, std::array<int, 3> { { 0, 0, 0 } }); } }
  static int class_of (int kind) { return kind == F_CELL ? 0 : kind == F_FC || kind == F_CHILD ? 1 : 2; }
  // room on level L for a0 / a1 / a2 more operations of the three classes
  bool room (int L, int a0, int a1, int a2) const
  {
    int c0 = a0, c1 = a1, c2 = a2;
    if (L >= 1 && L <= (int) cls.size ()) { c0 += cls[L - 1][0]; c1 += cls[L - 1][1]; c2 += cls[L - 1][2]; }
    if (!padded) return c0 + c1 + c2 <= width;
    return (c0 + 63)/64*64 + (c1 + 63)/64*64 + c2 <= width;
  }
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
    /* level 0: the values before the loop; a value at most FLOW_PD + 1 levels old is still in the LDS */
    return wlev[g] > 0 && L - wlev[g] <= FLOW_PD + 1 ? FLOW_LDS_REF (slot (wlev[g], wop[g])) : 8*g;
  }
  int emit (int L, const FlowRec & r)
  {
    need (L);
    lev[L - 1].push_back (r);
    cls[L - 1][class_of (r.w0 & 7)]++;
    return (int) lev[L - 1].size () - 1;
  }
  static FlowRec blank (int kind)
  {
    FlowRec r;
    r.w0 = kind; r.out_g = -1;
    for (int j = 0; j < FLOW_NIN; j++) r.in[j] = FLOW_NONE;
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
    for (int tries = 0; !room (L, 0, 0, 1); L++) if (++tries > 1000000) { ok = false; return; }
    FlowRec r = blank (F_GHOST);
    r.w0 |= (unsigned) constant (s) << 11;
    r.out_g = 8*g;
    r.in[0] = ref (img, L);
    const int i = emit (L, r);
    wlev[g] = L; wop[g] = i; rlev[g] = 0;
  }
  void cell (int g, int self, const FlowFace * f, int cell_level, bool reads_self)
  {
    int L = std::max (rlev[g], wlev[g]) + 1;
    int nfc = 0, nsum = 0, n2 = 0;      /* FC and SUM one level before the cell's, CHILD two levels before */
    if (reads_self) L = std::max (L, wlev[self] + 1);
    for (int d = 0; d < 2*dim; d++) {
      const FlowFace & F = f[d];
      if (F.kind == K_SAME) L = std::max (L, wlev[F.g] + 1);
      else if (F.kind == K_FC) { L = std::max (L, ready (F.fc) + 1); nfc++; }
      else if (F.kind == K_DEEP) {
	for (int i = 0; i < F.nch; i++) { L = std::max (L, ready (F.ch[i]) + 2); n2++; }
	nsum++;
      }
    }
    if (n2) L = std::max (L, 3); else if (nfc + nsum) L = std::max (L, 2);
    for (int tries = 0; !room (L, 1, 0, 0) || ((nfc || nsum) && !room (L - 1, 0, nfc, nsum)) || (n2 && !room (L - 2, 0, n2, 0)); L++)
      if (++tries > 1000000) { ok = false; return; }
    FlowRec r = blank (F_CELL);
    r.w0 |= (unsigned) cell_level << 15;
    r.out_g = 8*g;
    if (reads_self) { r.w0 |= 1u << 20; r.in[0] = ref (self, L); }
    for (int d = 0; d < 2*dim; d++) {
      const FlowFace & F = f[d];
      if (F.kind == K_SAME) {
	r.w0 |= FK_SAME << (3 + 2*d);
	r.in[1 + d] = ref (F.g, L);
      }
      else if (F.kind == K_FC) {
	r.w0 |= FK_PAIR << (3 + 2*d);
	r.in[1 + d] = FLOW_LDS_REF (slot (L - 1, emit_item (F_FC, F.fc, L - 1)));
      }
      else if (F.kind == K_DEEP) {
	FlowRec s = blank (F_SUM);
	s.w0 |= F.nch << 3;
	for (int i = 0; i < F.nch; i++)
	  s.in[i] = FLOW_LDS_REF (slot (L - 2, emit_item (F_CHILD, F.ch[i], L - 2)));
	r.w0 |= FK_PAIR << (3 + 2*d);
	r.in[1 + d] = FLOW_LDS_REF (slot (L - 1, emit (L - 1, s)));
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
		       bool reads_self, FlowPlan * out, bool host_only, int width = FLOW_WIDTH)
{
  flow_free (*out);
  FlowBuilder B;
  B.dim = dim;
  B.width = width;
  B.padded = width >= 256;      /* narrower plans (lab) pack the kinds: three kinds at multiples of 64 need the room */
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
  if (!B.ok || (int) B.lev.size () > FLOW_MAXLEV) return false;
  int nmixed_out = 0;
  {
    // the operations of a level sorted by kind (stable); the inputs that name an operation of the level by its
    // place follow.  An input of level L in buffer b was produced by the level L' in [L - FLOW_PD - 1, L - 1]
    // with (L' - 1) % FLOW_NBUF == b (FLOW_NBUF = FLOW_PD + 2: one such level).
    static const int order_of[8] = { 0, 1, 1, 2, 3, 4, 4, 4 };       /* CELL, FC / CHILD, SUM, GHOST */
    const int nl = (int) B.lev.size ();
    std::vector<std::vector<int>> place (nl);      /* place[L - 1][old] = new */
    int nmixed = 0;      /* levels where CELL and FC / CHILD (or CELL and the short kinds) share a wavefront */
    for (int L = 1; L <= nl; L++) {
      std::vector<FlowRec> & ops = B.lev[L - 1];
      std::vector<int> idx (ops.size ());
      for (size_t i = 0; i < idx.size (); i++) idx[i] = (int) i;
      std::stable_sort (idx.begin (), idx.end (), [&] (int a, int b) {
	  return order_of[ops[a].w0 & 7] < order_of[ops[b].w0 & 7]; });
      // where the level has room, a kind starts at a multiple of 64: a wavefront with two kinds runs the
      // branchy arithmetic, twice the time of the others, and the level waits for it.  CELL | FC, CHILD |
      // SUM, GHOST (the last two are short and share)
      int count[3] = { 0, 0, 0 };
      for (const FlowRec & r : ops) count[std::min (order_of[r.w0 & 7], 2)]++;
      int base[3] = { 0, count[0], count[0] + count[1] };
      {
	const int b1 = (count[0] + 63)/64*64, b2a = (b1 + count[1] + 63)/64*64, b2b = (count[0] + count[1] + 63)/64*64;
	if (count[1] + count[2] > 0 && b1 + count[1] + count[2] <= width) {
	  base[1] = b1;
	  base[2] = count[2] > 0 && b2a + count[2] <= width ? b2a : b1 + count[1];
	}
	else if (count[2] > 0 && b2b + count[2] <= width)
	  base[2] = b2b;
      }
      const int total = std::max (base[2] + count[2], std::max (base[1] + count[1], count[0]));
      if ((count[1] > 0 && base[1] % 64) || (count[2] > 0 && count[0] + count[1] > 0 && base[2] % 64 && count[1] == 0)) nmixed++;
      place[L - 1].resize (ops.size ());
      std::vector<FlowRec> sorted (total, FlowBuilder::blank (F_NOP));
      int rank[3] = { 0, 0, 0 };
      for (size_t i = 0; i < idx.size (); i++) {
	const int k = std::min (order_of[ops[idx[i]].w0 & 7], 2);
	const int at = base[k] + rank[k]++;
	sorted[at] = ops[idx[i]];
	place[L - 1][idx[i]] = at;
      }
      ops.swap (sorted);
    }
    for (int L = 1; L <= nl; L++)
      for (FlowRec & r : B.lev[L - 1])
	for (int j = 0; j < FLOW_NIN; j++)
	  if (FLOW_IS_LDS (r.in[j]) && r.in[j] != FLOW_NONE) {
	    const int sl = FLOW_LDS_ADDR (r.in[j]) >> 4, b = sl/width, i = sl % width;
	    int Lp = -1;
	    for (int q = L - 1; q >= std::max (1, L - FLOW_PD - 1); q--)
	      if ((q - 1) % FLOW_NBUF == b) Lp = q;
	    if (Lp < 0 || i >= (int) place[Lp - 1].size ()) return false;
	    r.in[j] = FLOW_LDS_REF (b*width + place[Lp - 1][i]);
	  }
    nmixed_out = nmixed;
  }
  FlowPlan & F = *out;
  F.width = width;
  if (getenv ("GFSHIP_TREE_DEBUG")) fprintf (stderr, "gfship_tree: flow plan: %d of %zu levels have a wavefront with two kinds\n", nmixed_out, B.lev.size ());
  {
    // places in the order of the plan
    std::vector<int> pos (ncell, -1);
    std::vector<int> & gidx = F.h_gidx;
    gidx.clear ();
    auto place_of = [&] (int g) { if (pos[g] < 0) { pos[g] = (int) gidx.size (); gidx.push_back (g); } return pos[g]; };
    for (auto & l : B.lev) for (FlowRec & r : l) if (r.out_g >= 0) place_of (r.out_g/8);
    for (auto & l : B.lev) for (FlowRec & r : l)
      for (int j = 0; j < FLOW_NIN; j++) if (!FLOW_IS_LDS (r.in[j])) place_of (r.in[j]/8);
    for (auto & l : B.lev) for (FlowRec & r : l) {
      if (r.out_g >= 0) r.out_g = 8*pos[r.out_g/8];
      for (int j = 0; j < FLOW_NIN; j++) if (!FLOW_IS_LDS (r.in[j])) r.in[j] = 8*pos[r.in[j]/8];
    }
    if (gidx.empty ()) gidx.push_back (0);
    F.npos = (int) gidx.size ();
    while (gidx.size () < FLOW_WIDTH + 64) gidx.push_back (gidx[0]);      /* spare places the idle lanes load (not unpacked) */
  }
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
  for (int k = 0; k < FLOW_WIDTH; k++) rec.push_back (FlowBuilder::blank (F_NOP));      /* what the idle lanes read */
  F.nct = (int) B.ct.size ();
  if (host_only) { F.h_rec = rec; F.h_lev_off = off; F.h_ct = B.ct; return true; }
  const std::vector<int> gidx = F.h_gidx;
  F.h_gidx.clear (); F.h_gidx.shrink_to_fit ();
  if (rec.empty ()) return false;
  std::vector<double> ct = B.ct;
  if (ct.empty ()) ct.push_back (0.);
  if (hipMalloc ((void **) &F.rec, rec.size ()*sizeof (FlowRec)) != hipSuccess ||
      hipMalloc ((void **) &F.lev_off, off.size ()*sizeof (int)) != hipSuccess ||
      hipMalloc ((void **) &F.ct, ct.size ()*sizeof (double)) != hipSuccess ||
      hipMalloc ((void **) &F.gidx, gidx.size ()*sizeof (int)) != hipSuccess ||
      hipMalloc ((void **) &F.up, gidx.size ()*sizeof (double)) != hipSuccess ||
      hipMalloc ((void **) &F.rp, gidx.size ()*sizeof (double)) != hipSuccess ||
      hipMemcpy (F.gidx, gidx.data (), gidx.size ()*sizeof (int), hipMemcpyHostToDevice) != hipSuccess ||
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
  std::vector<double> & ucells = u;
  std::vector<double> up (F.npos), rp (F.npos);
  for (int p = 0; p < F.npos; p++) { up[p] = ucells[F.h_gidx[p]]; rp[p] = rhs[F.h_gidx[p]]; }
  std::vector<FlowPair> lo (FLOW_LDS_SLOTS, FlowPair { 0., 0. });
  std::vector<std::vector<double>> pre (FLOW_NBUF, std::vector<double> ((size_t) FLOW_WIDTH*(FLOW_NIN + 1)));
  std::vector<int> stored_at (F.npos, -10);
  auto prefetch = [&] (int L, std::vector<double> & v) {      /* what flow_prefetch loads for level L (0-based) */
    if (L >= F.nlev) return;
    for (int i = F.h_lev_off[L]; i < F.h_lev_off[L + 1]; i++) {
      const FlowRec & r = F.h_rec[i];
      double * vi = &v[(size_t) (i - F.h_lev_off[L])*(FLOW_NIN + 1)];
      for (int j = 0; j < FLOW_NIN; j++)
	vi[j] = up[std::max (r.in[j], 0)/8];
      vi[FLOW_NIN] = rp[std::max (r.out_g, 0)/8];
    }
  };
  for (int L = 0; L < FLOW_PD; L++) prefetch (L, pre[L % FLOW_NBUF]);
  std::vector<std::pair<int, double>> late;      /* the stores of the level before: issued during this one */
  for (int L = 0; L < F.nlev; L++) {
    prefetch (L + FLOW_PD, pre[(L + FLOW_PD) % FLOW_NBUF]);
    for (auto & st : late) {
      if (stored_at[st.first] == L - 1) hazards++;      /* two stores of one level onto one cell */
      up[st.first] = st.second;
      stored_at[st.first] = L - 1;
    }
    late.clear ();
    std::vector<std::pair<int, double>> stores;
    std::vector<FlowPair> cur (FLOW_WIDTH, FlowPair { 0., 0. });
    for (int i = F.h_lev_off[L + 1]; i-- > F.h_lev_off[L]; ) {
      const FlowRec & r = F.h_rec[i];
      const double * vi = &pre[L % FLOW_NBUF][(size_t) (i - F.h_lev_off[L])*(FLOW_NIN + 1)];
      double x[FLOW_NIN], y[FLOW_NIN];
      for (int j = 0; j < FLOW_NIN; j++) {
	const int sl = FLOW_LDS_ADDR (r.in[j]) >> 4;
	if (FLOW_IS_LDS (r.in[j]) && r.in[j] != FLOW_NONE && sl/F.width == L % FLOW_NBUF)
	  hazards++;       /* an input in the buffer this level writes */
	const FlowPair p = lo[sl];
	x[j] = FLOW_IS_LDS (r.in[j]) ? p.x : vi[j];
	y[j] = p.y;
      }
      // the wavefront of this operation: the 64 operations around it (flow_eval_wave)
      const int w0i = F.h_lev_off[L] + (i - F.h_lev_off[L])/64*64, w1i = std::min (w0i + 64, F.h_lev_off[L + 1]);
      int first = F_NOP;      /* the kind of the first operation that is not a NOP, as flow_eval_wave takes it */
      for (int q = w0i; q < w1i && first == F_NOP; q++) first = flow_class (F.h_rec[q].w0 & 7);
      bool uniform = true, any3 = false;
      for (int q = w0i; q < w1i; q++) {
	const unsigned qw = F.h_rec[q].w0;
	if (flow_class (qw & 7) != first && flow_class (qw & 7) != F_NOP) uniform = false;
	if (flow_class (qw & 7) == F_FC)
	  for (int t = 0; t < dim - 1; t++)
	    if (t < (int) ((qw >> 3) & 3) && ((qw >> (5 + 3*t)) & 7) == 3) any3 = true;
      }
      struct AnyHost { bool v; bool operator() (bool) const { return v; } } any = { any3 };
      FlowPair o = { 0., 0. };
#define FLOW_EMU(D)							\
      o = !uniform ? flow_eval<D> (r.w0, x, y, vi[FLOW_NIN], F.h_ct.data (), omega, op, w) : \
	first == F_CELL ? flow_eval_uniform<D, F_CELL> (r.w0, x, y, vi[FLOW_NIN], F.h_ct.data (), omega, op, w, any) : \
	first == F_FC ? flow_eval_uniform<D, F_FC> (r.w0, x, y, vi[FLOW_NIN], F.h_ct.data (), omega, op, w, any) : \
	first == F_SUM ? flow_eval_uniform<D, F_SUM> (r.w0, x, y, vi[FLOW_NIN], F.h_ct.data (), omega, op, w, any) : \
	flow_eval_uniform<D, F_GHOST> (r.w0, x, y, vi[FLOW_NIN], F.h_ct.data (), omega, op, w, any)
      if ((r.w0 & 7) != F_NOP) { if (dim == 3) { FLOW_EMU (3); } else { FLOW_EMU (2); } }
#undef FLOW_EMU
      cur[i - F.h_lev_off[L]] = o;
      if (r.out_g >= 0) stores.push_back ({ r.out_g/8, o.x });
      for (int j = 0; j < FLOW_NIN; j++)
	if (!FLOW_IS_LDS (r.in[j]) && stored_at[r.in[j]/8] >= L - FLOW_PD - 1)
	  hazards++;         /* the load was issued in level L - FLOW_PD, beside the stores of L - FLOW_PD - 1 */
    }
    late = stores;
    for (int i = 0; i < F.width; i++) lo[(size_t) (L % FLOW_NBUF)*F.width + i] = cur[i];
  }
  for (auto & st : late) up[st.first] = st.second;
  for (int p = 0; p < F.npos; p++) ucells[F.h_gidx[p]] = up[p];
  return hazards;
}
