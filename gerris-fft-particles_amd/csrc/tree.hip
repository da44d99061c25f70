// tree.hip -- the reference's time step on a statically refined quadtree (2-D) or octree (3-D) in
// one periodic box (the coarse-fine stencils; the case of test/periodic/periodic.gfs
// with BOX = 1, 2 and its 3-D analogues).
//
// How the tree algorithms of the reference map onto the device:
//  * cell loops whose result does not depend on the order (src/ftt.c:689-926 traversals of leaves,
//    of the non-leaf cells of a level ...) are one thread per entry of a list built once on the host;
//  * face loops that scatter into both cells of a face (ftt_face_traverse, src/ftt.c:2152-2215:
//    MAC velocities, pressure correction, fluxes) are split into a kernel that computes one number
//    per face and a kernel in which every cell gathers the numbers of its faces IN THE ORDER the
//    reference's traversal visits them -- the same floating-point sums, no atomics;
//  * the Gauss-Seidel sweep of gfs_relax visits the cells of a level and the coarser leaves in tree
//    order (src/poisson.c:604-632): the host derives, from the very stencil code the kernels run
//    (tree.hpp with a recording reader), which cells each cell reads, and groups the cells of the
//    sweep into dependency levels; one workgroup then runs a whole relax loop (nrelax sweeps with
//    the periodic copies between them, src/poisson.c:1070-1089), level after level, barrier between
//    -- by default as a dataflow program of fixed-format micro-operations (tree_flow.hpp, t_relax_flow),
//    else from the tapes of the compiled stencils (t_relax_nodes_pf, t_relax_tape) or by the code that
//    walks the tree (t_relax_loop).
// The 2-D refined cases are small (10^4 - 10^5 cells): this path is about the reference's results on
// a tree, not about bandwidth; the uniform 3-D path of the other files is the one that is benchmarked.
//
// Restated here: src/poisson.c:998-1269 (cycle, solve), src/timestep.c:36-187,356-444,498-530,560-596,
// 644-717,872-921,976-1016, src/advection.c:27-99,132-180,267-343,398-435,513-587,
// src/fluid.c:1843-1864,2310-2324, src/domain.c:2239-2288,2824-2923, src/simulation.c:432-557,
// 1569-1633, src/ftt.c:45-83,169-192,2013-2074 (refinement with the neighbour and corner rules).
#include "gfship_internal.hpp"
#include "tree.hpp"
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <new>

using namespace gfship;
using namespace gfship::tree;

namespace {

enum { V_P, V_PMAC, V_U /* 3 */, V_G = V_U + 3 /* 3 */, V_GM = V_G + 3 /* 3 */, V_UN = V_GM + 3 /* 6 */,
       V_FV = V_UN + 6 /* 6 */, V_DIV = V_FV + 6, V_RES, V_DP, V_BCVAL, V_T /* 2: GfsVariableTracer */,
       V_BCU = V_T + 2 /* 3: the values of the conditions of U, V, W per ghost cell */,
       V_DRHS = V_BCU + 3 /* right-hand side of the implicit diffusion */, V_NVAR };
#define TREE_MAXTRACERS 2

// the variables of the C ABI (GFSHIP_TREE_*) -> storage
const int abi_var[] = { V_P, V_PMAC, V_U, V_U + 1, V_G, V_G + 1, V_GM, V_GM + 1, V_UN, V_UN + 1, V_UN + 2,
			V_UN + 3, V_U + 2, V_G + 2, V_GM + 2, V_UN + 4, V_UN + 5, V_DIV, V_BCVAL, V_RES,
			V_T, V_T + 1, V_BCU, V_BCU + 1, V_BCU + 2 };
const int abi_nvar = sizeof (abi_var)/sizeof (abi_var[0]);

struct P3 { double * p[3]; };       // the components of a vector
struct P6 { double * p[6]; };       // a number per direction

struct FaceRec { Cell cell, neighbor; int d; };
struct Ghost { int g, img, side, l; };   // img: the periodic image, or the cell the ghost touches (GfsBoundary)
struct Sgn6 { double s[6]; };            // homogeneous condition of each side: ghost = s * cell

struct FaceSet {            // the faces of one ftt_face_traverse, in its order
  int nfaces = 0;
  FaceRec * faces = nullptr;          // device
  int * inc_off = nullptr;            // device, per leaf (position in the leaf list) + 1
  int * inc = nullptr;                // device: face << 1 | role (0: the cell of the face, 1: its neighbour)
  double * fval = nullptr;            // device, one number per face
};

struct FlowPlan;            // tree_flow.hpp: the relax loop as a dataflow program of micro-operations

struct Sweep {              // T_LEVEL_LEAFS (m) in dependency levels
  int ncells = 0, nlev = 0;
  Cell * cells = nullptr;             // device, sorted by level (stable: traversal order inside)
  int * lev_off = nullptr;            // device, nlev + 1
  Ghost * ghosts = nullptr; int nghosts = 0;   // ghost cells of the selection
  // the stencils of the sweep compiled once (the tree is static): per cell, in the order of `cells',
  // a stream of codes / counts (ti), of constant coefficients (td) and of the cells whose values it
  // reads (tv); the cells of a dependency level in chunks that fit the LDS (stencil_tape)
  int * ti = nullptr, * tv = nullptr;
  double * td = nullptr;
  int * cell_off = nullptr;           // device, 3*(ncells + 1): start of each cell in ti / td / tv
  int * chunk = nullptr; int nchunks = 0;   // device, 2*(nchunks): first cell, dependency level; + end marker
  bool taped = false;
  // host copies of the streams and of the order, for the plan of a whole relax loop (loop_plan)
  std::vector<int> h_ti, h_tv, h_cell_off, h_g;
  std::vector<double> h_td;
  std::vector<Ghost> h_ghosts;
  // a whole relax loop (nrelax sweeps + the copies of the ghosts between them) as ONE list of
  // nodes in dependency levels: sweep s + 1 of a cell starts as soon as what it reads is final
  struct Loop {
    unsigned nrelax = 0;
    int nnodes = 0, nlev = 0, nchunks = 0;
    int * ti = nullptr, * tv = nullptr, * node_off = nullptr, * node_g = nullptr, * chunk = nullptr;
    double * td = nullptr;
    // the same plan for t_relax_nodes_pf: per chunk { c0, c1, i0, i1, d0, d1, v0, v1 }, per node { start in ti, td,
    // tv, cell } -- one load each, issued a chunk ahead
    int * desc = nullptr, * rec = nullptr;
    // host copies (gfship_tree_host_check)
    std::vector<int> h_ti, h_tv, h_node_off, h_node_g, h_node_level;
    std::vector<double> h_td;
    double sg[6] = { 0., 0., 0., 0., 0., 0. };   // the homogeneous conditions compiled into the copies of the ghosts
    FlowPlan * flow = nullptr;                   // the same loop for t_relax_flow (nullptr: not expressible)
  } loop;
  // the loops of the diffusion solves (10 nrelax sweeps on the first level, the conditions of U, V, W)
  Loop dloop[6];
  int dloop_next = 0;
};

struct DevReader {
  const double * p;
  __device__ inline double operator() (const Topo & T, Cell c) const { return p[T.gi (c)]; }
};

struct Recorder {           // host: which cells does the stencil read?
  std::vector<int> * out;
  inline double operator() (const Topo & T, Cell c) const { out->push_back (T.gi (c)); return 1.; }
};

} // namespace

struct gfship_tree {
  int device = 0;
  hipStream_t stream = nullptr;
  Topo H, D;
  std::vector<unsigned char> hflag;
  unsigned char * dflag = nullptr;
  std::vector<int> h_nbtab, h_child0;
  std::vector<unsigned char> h_cmask, h_idtab;
  int * d_nbtab = nullptr, * d_child0 = nullptr;
  unsigned char * d_cmask = nullptr, * d_idtab = nullptr;
  int ncell = 0;
  double * var[V_NVAR] = {};
  int nleaves = 0;
  Cell * leaves = nullptr;                        // device: interior leaves in traversal order
  std::vector<Cell> hleaves;
  int nnonleaf[GFSHIP_MAXLEVEL + 1] = {};
  Cell * nonleaf[GFSHIP_MAXLEVEL + 1] = {};       // device: interior non-leaf cells of a level
  Ghost * ghost_leaves = nullptr; int nghost_leaves = 0;
  Sweep sweep[GFSHIP_MAXLEVEL + 1];
  FaceSet fs[4];                                  // 0: FTT_XYZ, 1 + c: the faces normal to c
  double * d_red = nullptr;                       // reductions: [0] max bits / min bits, [1..3] sums
  double * h_red = nullptr;                       // pinned
  gfship_multilevel_params projection_params, approx_projection_params;
  double cfl = 0.8, dt = 0., t = 0., end = DBL_MAX, tnext = 0.;
  unsigned iter = 0;
  gfship_next_event_fn next_event = nullptr; void * next_event_ctx = nullptr;
  int side[6] = { 0, 0, 0, 0, 0, 0 };             // GFSHIP_SIDE_PERIODIC / GFSHIP_SIDE_BOUNDARY
  int bc_p[6] = { 0, 0, 0, 0, 0, 0 };             // condition of P on a GfsBoundary side (GFSHIP_BC_*)
  bool has_boundary = false;
  bool tape_attr_set = false;                     // dynamic-LDS limit of t_relax_tape raised on this device
  int ntracers = 0, tracer_gradient[TREE_MAXTRACERS] = { 1, 1 };   // GfsVariableTracer: 0 centred, 1 van Leer
  // conditions of U, V, W on GfsBoundary sides (GFSHIP_BC_SYMMETRY: the default GfsBc) with their values in
  // V_BCU + c; GfsSourceDiffusion {} U|V|W nu with its GfsMultilevelParams
  int bc_u[3][6] = {};
  double visc[3] = { 0., 0., 0. };
  gfship_multilevel_params diffusion_params[3];
  double src[3] = { 0., 0., 0. };   // GfsSource {} U|V|W g: constant intensities
};

namespace {

// ---- kernels --------------------------------------------------------------------------------

__global__ void t_copy_ghosts (const Ghost * gh, int n, double * v)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t < n) v[gh[t].g] = v[gh[t].img];
}

// symmetry (the default GfsBc, src/boundary.c:45-62) of a vector component / periodic copy: ghost = s * cell
__global__ void t_copy_ghosts_signed (const Ghost * gh, int n, double * v, Sgn6 sg)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t < n) v[gh[t].g] = sg.s[gh[t].side]*v[gh[t].img];
}

// the conditions of P on GfsBoundary sides (src/boundary.c:45-62,253-279,336-347): symmetry (scalar),
// Dirichlet 2 val - nb, Neumann nb + val h; periodic sides: the copy
__global__ void t_bc_values (const Ghost * gh, int n, double * v, const double * bcval, Sgn6 kind, int comp = -1)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Ghost G = gh[t];
  const double nb = v[G.img];
  const int k = (int) kind.s[G.side];     /* -1 periodic, else GFSHIP_BC_* */
  double x = nb;
  if (k == GFSHIP_BC_SYMMETRY && comp == G.side/2)    /* the normal component of a vector, src/boundary.c:45-51 */
    x = - nb;
  else if (k == GFSHIP_BC_DIRICHLET)
    x = 2.*bcval[G.g] - nb;
  else if (k == GFSHIP_BC_NEUMANN)
    x = nb + bcval[G.g]*(1./(1 << G.l));
  v[G.g] = x;
}

// gfs_domain_face_bc on periodic sides (src/boundary.c:1251-1258,1343-1347): the leaf ghost beyond
// side sd takes f[OPP (sd)].v of its image
// wall: 0 periodic, 1 + GFSHIP_BC_* of the variable on a GfsBoundary side (face_dirichlet src/boundary.c:270-275,
// face_neumann :349-355 with the values of the condition per ghost cell)
__global__ void t_face_bc (const Ghost * gh, int n, P6 fv, Sgn6 wall, int comp, const double * v = nullptr,
			   const double * bcval = nullptr)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Ghost G = gh[t];
  double * a = fv.p[G.side ^ 1];
  if (wall.s[G.side] == 0.) {       /* periodic side */
    a[G.g] = a[G.img];
    return;
  }
  const int kind = (int) wall.s[G.side] - 1;
  if (kind == GFSHIP_BC_DIRICHLET) {
    a[G.g] = fv.p[G.side][G.img] = bcval[G.g];
    return;
  }
  if (kind == GFSHIP_BC_NEUMANN) {
    a[G.g] = v[G.img] + bcval[G.g]*(1./(1 << G.l))/2.;
    return;
  }
  // face_symmetry, src/boundary.c:64-74 (img: the cell the ghost touches)
  double * own = fv.p[G.side];
  if (comp == G.side/2)
    a[G.g] = own[G.img] = 0.;
  else
    a[G.g] = own[G.img];
}

// gfs_get_from_below_intensive (src/fluid.c:1843-1864, mode 0) / get_from_below_2D
// (src/poisson.c:1057-1068, mode 1) on the non-leaf cells of one level
__global__ void t_from_below (Topo T, const Cell * cells, int n, double * v, int mode)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Cell c = cells[t];
  double val = 0., sa = 0.;
  for (int k = 0; k < T.nc (); k++) {
    const Cell ch = T.child (c, k);
    if (exists (ch)) {
      if (mode == 0) {
	val += v[T.gi (ch)]*1.;
	sa += 1.;
      }
      else
	val += v[T.gi (ch)];
    }
  }
  // get_from_below_2D: the sum; get_from_below_3D: the sum/2 (src/poisson.c:1044-1068)
  v[T.gi (c)] = mode == 0 ? val/sa : T.dim == 2 ? val : val/2.;
}

// get_from_above, src/poisson.c:1005-1042
__global__ void t_from_above (Topo T, const Cell * cells, int n, double * v)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Cell parent = cells[t];
  DevReader R = { v };
  const double vp = v[T.gi (parent)];
  double h[3];
  for (int c = 0; c < T.dim; c++) {
    Face f;
    f.cell = parent;
    f.d = 2*c;
    f.neighbor = T.neighbor (parent, f.d);
    Grad2 g = face_gradient (T, f, R, parent.l);
    const double g1 = g.b - g.a*vp;
    f.d = 2*c + 1;
    f.neighbor = T.neighbor (parent, f.d);
    g = face_gradient (T, f, R, parent.l);
    const double g2 = g.b - g.a*vp;
    h[c] = (g1 - g2)/2.;
  }
  for (int k = 0; k < T.nc (); k++) {
    const Cell ch = T.child (parent, k);
    if (exists (ch)) {
      const double px = (k & 1) ? 0.25 : -0.25, py = (k & 2) ? -0.25 : 0.25,
	pz = (k & 4) ? -0.25 : 0.25;                 /* ftt_cell_relative_pos */
      double x = vp;
      x += px*h[0];
      x += py*h[1];
      if (T.dim == 3)
	x += pz*h[2];
      v[T.gi (ch)] = x;
    }
  }
}

// relax_loop, src/poisson.c:1070-1089, of one level: one workgroup, the cells of a sweep by
// dependency level
__global__ void __launch_bounds__(1024)
t_relax_loop (Topo T, const Cell * cells, const int * lev_off, int nlev, const Ghost * gh, int ngh,
	      double * u, const double * rhs, unsigned nrelax, double omega, int max_level, Sgn6 sg,
	      int op = 0, double w = 1.)
{
  DevReader R = { u };
  for (int t = threadIdx.x; t < ngh; t += blockDim.x)
    u[gh[t].g] = sg.s[gh[t].side]*u[gh[t].img];
  __syncthreads ();
  for (unsigned s = 0; s < nrelax; s++) {
    for (int L = 0; L < nlev; L++) {
      const int a = lev_off[L], b = lev_off[L + 1];
      for (int t = a + threadIdx.x; t < b; t += blockDim.x) {
	const Cell c = cells[t];
	const int g = T.gi (c);
	u[g] = op == 0 ? relax_cell (T, c, R, rhs[g], omega, max_level) :
	  diffusion_relax_cell (T, c, R, rhs[g], w, max_level);
      }
      __syncthreads ();
    }
    if (s + 1 < nrelax) {
      for (int t = threadIdx.x; t < ngh; t += blockDim.x)
	u[gh[t].g] = sg.s[gh[t].side]*u[gh[t].img];
      __syncthreads ();
    }
  }
}

// ---- compiled stencils ------------------------------------------------------------------------
// relax_cell / face_gradient (tree.hpp) walk the tree for every cell of every sweep: a chain of
// dependent loads (neighbour table, flags, children, values) that one thread pays in full, and the
// slowest cell of a dependency level -- a coarse leaf with fine neighbours reads ~40 values through
// two levels of interpolation -- sets the time of the level.  The tree does not change: the host
// walks it once per cell (stencil_tape, below) and writes down what the walk found: which values are
// read, with which constant coefficients, combined in which order.  The sweep kernel stages the
// streams of a chunk of cells in LDS together with the gathered values (one independent load per
// thread), then evaluates them from LDS: same operations in the same order as the template code.
enum { K_NONE = 0, K_SAME = 1, K_FC = 2, K_DEEP = 3, K_GHOST = 4 };

struct TapeCursor {          // the streams of a cell, the values gathered beforehand (LDS / host)
  const int * ti; const double * td; const double * tv;
  __host__ __device__ inline double val () { return *tv++; }
};
struct TapeCursorG {         // the same with the values read through the stream of cells (global memory)
  const int * ti; const double * td; const int * tvi; const double * u;
  __device__ inline double val () { return u[*tvi++]; }
};

// p.b of interpolate_1D1 / interpolate_2D1 from the streams: sum of a_j * P_j, P_j a value or the
// average of the children of a refined neighbour (average_neighbor_value)
template <class CUR>
__host__ __device__ inline double tape_interpolation (CUR & c)
{
  const int nt = *c.ti++;
  double pb = 0.;
  for (int t = 0; t < nt; t++) {
    const double a = *c.td++;
    const int cnt = *c.ti++;
    double P;
    if (cnt == 0)
      P = c.val ();
    else {
      double av = 0., n = 0.;
      for (int k = 0; k < cnt; k++) {
	n += 1.;
	av += 1.*(c.val ());
      }
      P = av/n;
    }
    pb += a*P;
  }
  return pb;
}

// the sums g.a, g.b of relax / residual_set over the faces of a cell (src/poisson.c:507-557,634-678)
// w: the weight every face carries (1. for the Poisson problem with alpha = NULL; the diffusion
// coefficient of a quadtree, face_gradient_w of tree.hpp) -- the products by 1. leave the bits alone
template <class CUR>
__host__ __device__ inline void tape_cell (CUR & c, int nd, int dim, int ncd, double & ga, double & gb,
					   double w = 1.)
{
  ga = 0.; gb = 0.;
  for (int d = 0; d < nd; d++) {
    const int kind = *c.ti++;
    if (kind == K_NONE)
      continue;
    double na, nb;
    if (kind == K_SAME) {
      na = w;
      nb = w*(c.val ());
    }
    else if (kind == K_FC) {      /* gradient_fine_coarse towards a coarser neighbour */
      const double cb = *c.td++;
      const double uN = c.val ();
      const double pb = tape_interpolation (c);
      const double gc = 2.*pb/3.;
      na = w*(2./3.);
      nb = w*(cb*uN + gc);
    }
    else {                        /* the fine cells behind a face of a coarser leaf */
      const int nch = *c.ti++;
      na = 0.; nb = 0.;
      for (int i = 0; i < nch; i++) {
	const double gbi = *c.td++;
	const double uch = c.val ();
	const double pb = tape_interpolation (c);
	const double gc = 2.*pb/3.;
	na += w*gbi;
	nb += w*((2./3.)*uch - gc);
      }
      if (dim > 2) {
	na /= ncd/2.;
	nb /= ncd/2.;
      }
    }
    ga += na;
    gb += nb;
  }
}

#include "tree_flow.hpp"

#define TAPE_LDS_BYTES (120*1024)

// relax_loop (src/poisson.c:1070-1089) of one level from the compiled stencils: one workgroup; per
// chunk of a dependency level: stage the streams and gather the values (all threads), barrier,
// one thread per cell evaluates from LDS and stores the new value, barrier
__global__ void __launch_bounds__(1024)
t_relax_tape (Topo T, const Cell * cells, const int * cell_off, int ncells, const int * chunk, int nchunks,
	      const int * ti_g, const double * td_g, const int * tv_g,
	      const Ghost * gh, int ngh, double * u, const double * rhs, unsigned nrelax, double omega, Sgn6 sg)
{
  extern __shared__ double lds[];
  const int nd = T.nd (), dim = T.dim, ncd = T.ncd ();
  for (int t = threadIdx.x; t < ngh; t += blockDim.x)
    u[gh[t].g] = sg.s[gh[t].side]*u[gh[t].img];
  __syncthreads ();
  for (unsigned s = 0; s < nrelax; s++) {
    for (int k = 0; k < nchunks; k++) {
      const int c0 = chunk[k], c1 = chunk[k + 1];
      const int i0 = cell_off[3*c0], i1 = cell_off[3*c1];
      const int d0 = cell_off[3*c0 + 1], d1 = cell_off[3*c1 + 1];
      const int v0 = cell_off[3*c0 + 2], v1 = cell_off[3*c1 + 2];
      // LDS: values, then constants, then codes
      double * lv = lds, * ld = lds + (v1 - v0);
      int * li = (int *) (ld + (d1 - d0));
      for (int t = threadIdx.x; t < v1 - v0; t += blockDim.x)
	lv[t] = u[tv_g[v0 + t]];
      for (int t = threadIdx.x; t < d1 - d0; t += blockDim.x)
	ld[t] = td_g[d0 + t];
      for (int t = threadIdx.x; t < i1 - i0; t += blockDim.x)
	li[t] = ti_g[i0 + t];
      __syncthreads ();
      for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
	TapeCursor cur = { li + (cell_off[3*c] - i0), ld + (cell_off[3*c + 1] - d0), lv + (cell_off[3*c + 2] - v0) };
	const double self = *cur.tv++;
	double ga, gb;
	tape_cell (cur, nd, dim, ncd, ga, gb);
	const int g = T.gi (cells[c]);
	double x = 0.;
	if (ga != 0.)
	  x = dim == 2 ? (1. - omega)*self + omega*(gb - rhs[g])/ga : (gb - rhs[g])/ga;
	u[g] = x;
      }
      __syncthreads ();
    }
    if (s + 1 < nrelax) {
      for (int t = threadIdx.x; t < ngh; t += blockDim.x)
	u[gh[t].g] = sg.s[gh[t].side]*u[gh[t].img];
      __syncthreads ();
    }
  }
}

// the same for the plan of a whole relax loop (loop_plan): the nodes -- (cell, sweep) and the copies
// of the ghost cells between the sweeps -- by dependency level; the streams are laid out in node order
__global__ void __launch_bounds__(1024)
t_relax_nodes (Topo T, const int * node_g, const int * node_off, const int * chunk, int nchunks,
	       const int * ti_g, const double * td_g, const int * tv_g, double * u, const double * rhs,
	       double omega, int op = 0, double w = 1.)
{
  extern __shared__ double lds[];
  const int nd = T.nd (), dim = T.dim, ncd = T.ncd ();
  for (int k = 0; k < nchunks; k++) {
    const int c0 = chunk[k], c1 = chunk[k + 1];
    const int i0 = node_off[3*c0], i1 = node_off[3*c1];
    const int d0 = node_off[3*c0 + 1], d1 = node_off[3*c1 + 1];
    const int v0 = node_off[3*c0 + 2], v1 = node_off[3*c1 + 2];
    double * lv = lds, * ld = lds + (v1 - v0);
    int * li = (int *) (ld + (d1 - d0));
    for (int t = threadIdx.x; t < v1 - v0; t += blockDim.x)
      lv[t] = u[tv_g[v0 + t]];
    for (int t = threadIdx.x; t < d1 - d0; t += blockDim.x)
      ld[t] = td_g[d0 + t];
    for (int t = threadIdx.x; t < i1 - i0; t += blockDim.x)
      li[t] = ti_g[i0 + t];
    __syncthreads ();
    for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
      TapeCursor cur = { li + (node_off[3*c] - i0), ld + (node_off[3*c + 1] - d0), lv + (node_off[3*c + 2] - v0) };
      const int g = node_g[c];
      if (*cur.ti == K_GHOST)       /* homogeneous condition / periodic copy: ghost = s * cell */
	u[g] = (*cur.td)*(*cur.tv);
      else if (op == 0) {
	const double self = *cur.tv++;
	double ga, gb;
	tape_cell (cur, nd, dim, ncd, ga, gb);
	double x = 0.;
	if (ga != 0.)
	  x = dim == 2 ? (1. - omega)*self + omega*(gb - rhs[g])/ga : (gb - rhs[g])/ga;
	u[g] = x;
      }
      else {      /* diffusion_relax, src/poisson.c:1455-1484 (rhoc = 1): diffusion_relax_cell of tree.hpp */
	cur.tv++;
	double ga, gb;
	tape_cell (cur, nd, dim, ncd, ga, gb, w);
	int l = 0;
	while (g >= T.off[l + 1]) l++;
	const double h = 1./(1 << l);
	const double a = 1.*h*h;
	ga = 1. + ga/a;
	u[g] = (gb/a + rhs[g])/ga;
      }
    }
    __syncthreads ();
  }
}

// t_relax_nodes with everything that does not depend on the values loaded a chunk ahead.  A step of
// t_relax_nodes is a chain of seven dependent global round trips (chunk bounds, node offsets, gather index,
// value, node cell, right-hand side, store acknowledgement: 2.9 us); here the descriptor of the next chunk is
// loaded while this one is staged, the node record and the gather indices of the next chunk while this one is
// evaluated, its right-hand side behind the stores: what is left on the chain is the gather of the values,
// the evaluation from LDS and the acknowledgement of the stores.
#define PF_R 8      /* gather indices a thread keeps for the next chunk (chunks with more: loaded in place) */
struct I8 { int v[8]; };

__global__ void __launch_bounds__(1024)
t_relax_nodes_pf (Topo T, const int * rec_g, const int * desc_g, int nchunks,
		  const int * ti_g, const double * td_g, const int * tv_g, double * u, const double * rhs,
		  double omega, int op, double w, const int * node_g, const int * node_off)
{
  extern __shared__ double lds[];
  const int nd = T.nd (), dim = T.dim, ncd = T.ncd ();
  const int tid = threadIdx.x;
  typedef int int4v __attribute__((ext_vector_type(4)));
  I8 D = *(const I8 *) desc_g;
  int4v rec = { 0, 0, 0, 0 };
  double rhs_r = 0.;
  int idx[PF_R];
  if (D.v[0] + tid < D.v[1]) {
    rec = *(const int4v *) (rec_g + 4*(size_t) (D.v[0] + tid));
    rhs_r = rhs[rec.w];
  }
#pragma unroll
  for (int r = 0; r < PF_R; r++) {
    const int t = tid + r*1024;
    idx[r] = t < D.v[7] - D.v[6] ? tv_g[D.v[6] + t] : 0;
  }
  for (int k = 0; k < nchunks; k++) {
    const int c0 = D.v[0], c1 = D.v[1], i0 = D.v[2], i1 = D.v[3], d0 = D.v[4], d1 = D.v[5], v0 = D.v[6], v1 = D.v[7];
    double * lv = lds, * ld = lds + (v1 - v0);
    int * li = (int *) (ld + (d1 - d0));
    // ---- stage: the values (indices already here), the constants, the codes
#pragma unroll
    for (int r = 0; r < PF_R; r++) {
      const int t = tid + r*1024;
      if (t < v1 - v0) lv[t] = u[idx[r]];
    }
    for (int t = tid + PF_R*1024; t < v1 - v0; t += 1024)
      lv[t] = u[tv_g[v0 + t]];
    for (int t = tid; t < d1 - d0; t += 1024)
      ld[t] = td_g[d0 + t];
    for (int t = tid; t < i1 - i0; t += 1024)
      li[t] = ti_g[i0 + t];
    const I8 Dn = *(const I8 *) (desc_g + 8*(size_t) (k + 1));      /* zeros behind the last chunk */
    __syncthreads ();
    // ---- the loads for the next chunk that do not depend on this one's results
    int4v recn = { 0, 0, 0, 0 };
    int idxn[PF_R];
    const bool nextnode = Dn.v[0] + tid < Dn.v[1];
    if (nextnode)
      recn = *(const int4v *) (rec_g + 4*(size_t) (Dn.v[0] + tid));
#pragma unroll
    for (int r = 0; r < PF_R; r++) {
      const int t = tid + r*1024;
      idxn[r] = t < Dn.v[7] - Dn.v[6] ? tv_g[Dn.v[6] + t] : 0;
    }
    // ---- evaluate
    for (int c = c0 + tid; c < c1; c += 1024) {
      int io, dof, vo, g;
      double rh;
      if (c == c0 + tid) { io = rec.x; dof = rec.y; vo = rec.z; g = rec.w; rh = rhs_r; }
      else { io = node_off[3*c]; dof = node_off[3*c + 1]; vo = node_off[3*c + 2]; g = node_g[c]; rh = rhs[g]; }
      TapeCursor cur = { li + (io - i0), ld + (dof - d0), lv + (vo - v0) };
      if (*cur.ti == K_GHOST)       /* homogeneous condition / periodic copy: ghost = s * cell */
	u[g] = (*cur.td)*(*cur.tv);
      else if (op == 0) {
	const double self = *cur.tv++;
	double ga, gb;
	tape_cell (cur, nd, dim, ncd, ga, gb);
	double x = 0.;
	if (ga != 0.)
	  x = dim == 2 ? (1. - omega)*self + omega*(gb - rh)/ga : (gb - rh)/ga;
	u[g] = x;
      }
      else {      /* diffusion_relax, src/poisson.c:1455-1484 (rhoc = 1) */
	cur.tv++;
	double ga, gb;
	tape_cell (cur, nd, dim, ncd, ga, gb, w);
	int l = 0;
	while (g >= T.off[l + 1]) l++;
	const double h = 1./(1 << l);
	const double a = 1.*h*h;
	ga = 1. + ga/a;
	u[g] = (gb/a + rh)/ga;
      }
    }
    double rhs_n = 0.;
    if (nextnode)
      rhs_n = rhs[recn.w];
    __syncthreads ();
    D = Dn; rec = recn; rhs_r = rhs_n;
#pragma unroll
    for (int r = 0; r < PF_R; r++) idx[r] = idxn[r];
  }
}

__device__ inline void atomic_max_pos (double * addr, double v)   /* v >= 0 */
{
  atomicMax ((unsigned long long *) addr, (unsigned long long) __double_as_longlong (v));
}
__device__ inline void atomic_min_pos (double * addr, double v)
{
  atomicMin ((unsigned long long *) addr, (unsigned long long) __double_as_longlong (v));
}

// the sum / maximum / minimum of a wavefront in lane 0 (every lane of the wavefront calls these): one atomic
// per wavefront and number instead of one per cell -- 40 000 atomics on four addresses were the time of
// t_residual (80 us)
__device__ inline double wave_sum (double v)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down (v, o);
  return v;
}
__device__ inline double wave_max (double v)
{
  for (int o = 32; o > 0; o >>= 1) {      /* a NaN wins, as it does in atomic_max_pos (its bits are the largest) */
    const double b = __shfl_down (v, o);
    v = (v > b || v != v) ? v : b;
  }
  return v;
}
__device__ inline double wave_min (double v)
{
  for (int o = 32; o > 0; o >>= 1) v = fmin (v, __shfl_down (v, o));
  return v;
}

// gfs_residual on the leaves + add_norm_residual (src/domain.c:2239-2246): the maximum is exact, the
// sums are accumulated in no particular order (reported, never branched on)
__global__ void t_residual (Topo T, const Cell * cells, int n, const double * u, const double * rhs,
			    double * res, double * red)
{
  // (eight lanes per leaf, one per face, with the sums in direction order by lane 0: 74 -> 244 us -- the lanes of a
  // wavefront then walk eight different ways through the tree)
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  double r = 0., val = 0.;
  if (t < n) {
    const Cell c = cells[t];
    DevReader R = { u };
    const int g = T.gi (c);
    r = residual_cell (T, c, R, rhs[g]);
    res[g] = r;
    const double size = T.size (c);
    val = r/(1.*size*size);
  }
  const double m = wave_max (fabs (val)), s1 = wave_sum (r), s2 = wave_sum (fabs (val)), s3 = wave_sum (val*val);
  if ((threadIdx.x & 63) == 0) {
    atomic_max_pos (&red[0], m);
    atomicAdd (&red[1], s1);
    atomicAdd (&red[2], s2);
    atomicAdd (&red[3], s3);
  }
}

// the same from the compiled stencils of the finest sweep (its cells are the leaves): the streams of a cell say
// which values to read -- two dependent loads each instead of the chain of loads through the tree (tree.hpp)
// that made t_residual run at the latency of its longest thread
__global__ void t_residual_tape (Topo T, const Cell * cells, const int * cell_off, int n, const int * ti_g,
				 const double * td_g, const int * tv_g, const double * u, const double * rhs,
				 double * res, double * red)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  double r = 0., val = 0.;
  if (t < n) {
    TapeCursorG cur = { ti_g + cell_off[3*t], td_g + cell_off[3*t + 1], tv_g + cell_off[3*t + 2], u };
    const double self = cur.val ();
    double ga, gb;
    tape_cell (cur, T.nd (), T.dim, T.ncd (), ga, gb);
    const int g = T.gi (cells[t]);
    r = rhs[g] - (gb - self*ga);
    res[g] = r;
    const double size = T.size (cells[t]);
    val = r/(1.*size*size);
  }
  const double m = wave_max (fabs (val)), s1 = wave_sum (r), s2 = wave_sum (fabs (val)), s3 = wave_sum (val*val);
  if ((threadIdx.x & 63) == 0) {
    atomic_max_pos (&red[0], m);
    atomicAdd (&red[1], s1);
    atomicAdd (&red[2], s2);
    atomicAdd (&red[3], s3);
  }
}

__global__ void t_correct (Topo T, const Cell * cells, int n, double * u, const double * dp)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int g = T.gi (cells[t]);
  u[g] += dp[g];
}

// gfs_face_interpolated_normal_velocity, src/advection.c:549-573: the value of the face
// gfs_diffusion_rhs, src/poisson.c:1392-1451 (rhoc = 1, the face weight w of a quadtree)
__global__ void t_diffusion_rhs (Topo T, const Cell * cells, int n, const double * v, double * rhs, double w, double pbeta)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Cell c = cells[t];
  DevReader R = { v };
  rhs[T.gi (c)] += diffusion_rhs_cell (T, c, R, w, pbeta);
}

// gfs_diffusion_residual (src/poisson.c:1519-1556) on the leaves + gfs_domain_norm_variable
// (src/domain.c:2197-2232: weights = cell volumes): the maximum is exact, the sums in no particular order
__global__ void t_diffusion_residual (Topo T, const Cell * cells, int n, const double * u, const double * rhs,
				      double * res, double w, double * red)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  double r = 0., vol = 0.;
  if (t < n) {
    const Cell c = cells[t];
    DevReader R = { u };
    const int g = T.gi (c);
    r = diffusion_residual_cell (T, c, R, rhs[g], w);
    res[g] = r;
    const double size = T.size (c);
    vol = T.dim == 3 ? size*size*size : size*size;
  }
  const double m = wave_max (fabs (r)), s1 = wave_sum (vol*r), s2 = wave_sum (vol*fabs (r)), s3 = wave_sum (vol*r*r),
    s4 = wave_sum (vol);
  if ((threadIdx.x & 63) == 0) {
    atomic_max_pos (&red[0], m);
    atomicAdd (&red[1], s1);
    atomicAdd (&red[2], s2);
    atomicAdd (&red[3], s3);
    atomicAdd (&red[4], s4);
  }
}

__global__ void t_copy_leaves (Topo T, const Cell * cells, int n, const double * src, double * dst)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int g = T.gi (cells[t]);
  dst[g] = src[g];
}

__global__ void t_face_interp (Topo T, const FaceRec * faces, int n, P3 u, double * fval)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Face f = { faces[t].cell, faces[t].neighbor, faces[t].d };
  DevReader R = { u.p[f.d/2] };
  fval[t] = face_interpolated_value (T, f, R);
}

// the normal velocities of a leaf from the faces that touch it, in the order of the traversal:
// f[d].un of the cell of a face is set; that of its neighbour is set (same level) or gets half of
// the value (coarser neighbour, FTT_CELLS_DIRECTION = 2), after gfs_face_reset_normal_velocity
__global__ void t_gather_un (Topo T, const Cell * cells, int n, const FaceRec * faces, const int * inc_off,
			     const int * inc, const double * fval, P6 un, int dmask)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  double acc[6] = { 0., 0., 0., 0., 0., 0. };
  for (int k = inc_off[t]; k < inc_off[t + 1]; k++) {
    const int fi = inc[k] >> 1, role = inc[k] & 1;
    const FaceRec & f = faces[fi];
    const double u = fval[fi];
    if (role == 0)
      acc[f.d] = u;
    else if (f.neighbor.l == f.cell.l)
      acc[f.d ^ 1] = u;
    else
      acc[f.d ^ 1] += u*1./(1.*T.ncd ());
  }
  const int g = T.gi (cells[t]);
  for (int d = 0; d < T.nd (); d++)
    if (dmask & (1 << d))
      un.p[d][g] = acc[d];
}

// correct_normal_velocity, src/timestep.c:118-144: dp of the face
__global__ void t_face_correct (Topo T, const FaceRec * faces, int n, const double * p, double * fval)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Face f = { faces[t].cell, faces[t].neighbor, faces[t].d };
  DevReader R = { p };
  const Grad2 g = face_gradient (T, f, R, -1);
  double dp = (g.b - g.a*p[T.gi (f.cell)])/T.size (f.cell);
  if (f.d & 1)
    dp = - dp;
  dp /= 1.;
  fval[t] = dp;
}

__global__ void t_gather_correct (Topo T, const Cell * cells, int n, const FaceRec * faces,
				  const int * inc_off, const int * inc, const double * fval,
				  P6 un, P3 gv, double dt)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int g = T.gi (cells[t]);
  double gacc[3];
  for (int c = 0; c < 3; c++)
    gacc[c] = gv.p[c] ? gv.p[c][g] : 0.;
  for (int k = inc_off[t]; k < inc_off[t + 1]; k++) {
    const int fi = inc[k] >> 1, role = inc[k] & 1;
    const FaceRec & f = faces[fi];
    double dp = fval[fi];
    if (role == 0) {
      un.p[f.d][g] -= dp*dt;
      gacc[f.d/2] += dp*1.;
    }
    else {
      if (f.neighbor.l < f.cell.l)
	dp *= 1./(1.*T.nc ()/2);
      un.p[f.d ^ 1][g] -= dp*dt;
      gacc[f.d/2] += dp*1.;
    }
  }
  for (int c = 0; c < 3; c++)
    if (gv.p[c])
      gv.p[c][g] = gacc[c];
}

// gfs_normal_divergence + scale_divergence, src/fluid.c:2310-2324, src/timestep.c:181-187
__global__ void t_divergence (Topo T, const Cell * cells, int n, P6 un, double * div, double dt)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Cell c = cells[t];
  const int g = T.gi (c);
  double d = 0.;
  for (int e = 0; e < T.nd (); e++)
    d += ((e & 1) ? -1. : 1.)*un.p[e][g]*1.;
  d = d*T.size (c);
  div[g] = d/dt;
}

__global__ void t_scale (Topo T, const Cell * cells, int n, P3 a, double s, int mode, P3 g)
{
  // mode 0: a[c] /= 2 (scale_cell_gradients, src/timestep.c:60-90: both neighbours exist);
  // mode 1: a[c] -= g[c]*s (correct, src/timestep.c:486-496)
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int gi = T.gi (cells[t]);
  for (int c = 0; c < T.dim; c++) {
    if (mode == 0)
      a.p[c][gi] /= 2.;
    else
      a.p[c][gi] -= g.p[c][gi]*s;
  }
}

struct AdvArgs {
  const double * v, * u[3], * un[6];
  double * fv[6];
  double dt;
  int use_centered;
  int gradient;            // 0 gfs_center_gradient, 1 gfs_center_van_leer_gradient
  double visc;             // GfsSourceDiffusion on the variable: its explicit term is the MAC source (0: none)
  double gsrc;             // the intensity of a GfsSource on the variable (0: none)
};

// transverse_term, src/advection.c:27-47
__device__ inline double transverse_term (const Topo & T, const AdvArgs & A, Cell cell, int g, double v0,
					  DevReader & R, double msize, int ct)
{
  const double vtan = A.use_centered ? A.u[ct][g] : (A.un[2*ct][g] + A.un[2*ct + 1][g])/2.;
  Face f;
  f.d = vtan > 0. ? 2*ct + 1 : 2*ct;
  f.cell = cell;
  f.neighbor = T.neighbor (cell, f.d);
  const Grad2 gf = face_gradient (T, f, R, -1);
  double gt = gf.b - gf.a*v0;
  if (vtan > 0.) gt = - gt;
  return A.dt*vtan*gt/(2.*msize);
}

// gfs_cell_advected_face_values, src/advection.c:58-99 (centred gradient, no sources)
// one thread per leaf and direction: the three directions of a leaf are independent, and each is a chain of
// dependent loads through the tree (a leaf per thread left less than one wavefront per compute unit on the
// trees of the benches: the kernel ran at the latency of the longest chain)
__global__ void t_face_values (Topo T, const Cell * cells, int n, AdvArgs A)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n*T.dim) return;
  const Cell cell = cells[t/T.dim];
  const int g = T.gi (cell);
  DevReader R = { A.v };
  const double size = T.size (cell);
  const double v0 = A.v[g];
  {
    const int c = t % T.dim;
    const double msize = size;
    const double unorm = A.use_centered ? A.dt*A.u[c][g]/msize :
      A.dt*(A.un[2*c][g] + A.un[2*c + 1][g])/(2.*msize);
    const double gr = A.gradient ? van_leer_gradient (T, cell, c, R) : center_gradient (T, cell, c, R);
    const double m1 = (1. - unorm)/2., m2 = (- 1. - unorm)/2.;
    const double vl = v0 + (m1 < 0.5 ? m1 : 0.5)*gr;
    const double vr = v0 + (m2 > -0.5 ? m2 : -0.5)*gr;
    double msrc = 0.;        /* gfs_variable_mac_source, src/source.c:38-59 */
    if (A.visc != 0. || A.gsrc != 0.) {
      double sum = 0.;
      if (A.visc != 0.)
	sum += source_diffusion_value (T, cell, R, A.visc);
      if (A.gsrc != 0.)
	sum += A.gsrc;
      msrc = sum;
    }
    const double src = A.dt*msrc/2.;
    double dv;
    if (T.dim == 2)
      dv = transverse_term (T, A, cell, g, v0, R, msize, (c + 1) % 2);
    else {
      // orthogonal[c] = { {Y, Z}, {X, Z}, {X, Y} }
      const int o0 = c == 0 ? 1 : 0, o1 = c == 2 ? 1 : 2;
      dv =  transverse_term (T, A, cell, g, v0, R, msize, o0);
      dv += transverse_term (T, A, cell, g, v0, R, msize, o1);
    }
    A.fv[2*c][g]     = vl + src - dv;
    A.fv[2*c + 1][g] = vr + src - dv;
  }
}

struct UpwindArgs { const double * u[3], * un[6], * fv[6]; };

// interpolate_1D1 of src/advection.c:132-180 (the assigned values of s2: see oracle/go_tree.c)
__device__ inline double adv_interpolate_1D1 (const Topo & T, const UpwindArgs & A, Cell cell, int dright,
					      int dup, double x)
{
  const int dleft = dright ^ 1;
  Cell nb = T.neighbor (cell, dup);
  if (exists (nb) && T.interior (nb)) {
    double s2 = T.leaf (nb) ? 1. : 0.5;
    const double s1 = 1.;
    const double v1 = A.fv[dleft][T.gi (cell)];
    double v2;
    if (T.leaf (nb))
      v2 = A.fv[dleft][T.gi (nb)];
    else {
      // ftt_cell_child_corner: the child of nb in the corner (dleft, opposite of dup)
      nb = T.child_corner (nb, dleft, dup ^ 1, -1);
      if (exists (nb))
	v2 = A.fv[dleft][T.gi (nb)];
      else
	s2 = v2 = 0.;
    }
    return s2 > 0. ? (v2*(s1 - 1. + 2.*x) + v1*(s2 + 1. - 2.*x))/(s1 + s2) : v1;
  }
  return A.fv[dleft][T.gi (cell)];
}

// interpolate_2D1 of src/advection.c:183-249 (3-D)
__device__ inline double adv_interpolate_2D1 (const Topo & T, const UpwindArgs & A, Cell cell, int dright,
					      int d1, int d2, double x, double y)
{
  double x1 = 0., y1 = 1.;
  double x2 = 1., y2 = 0.;
  double v1, v2;
  const int dleft = dright ^ 1;
  const double v0 = A.fv[dleft][T.gi (cell)];
  Cell n1 = T.neighbor (cell, d1);
  if (exists (n1) && T.interior (n1)) {
    if (!T.leaf (n1)) {
      n1 = T.child_corner (n1, dright ^ 1, d1 ^ 1, d2);
      if (exists (n1)) {
	v1 = A.fv[dleft][T.gi (n1)];
	x1 = 1./4.;
	y1 = 3./4.;
      }
      else
	v1 = v0;
    }
    else
      v1 = A.fv[dleft][T.gi (n1)];
  }
  else
    v1 = v0;
  Cell n2 = T.neighbor (cell, d2);
  if (exists (n2) && T.interior (n2)) {
    if (!T.leaf (n2)) {
      n2 = T.child_corner (n2, dright ^ 1, d2 ^ 1, d1);
      if (exists (n2)) {
	v2 = A.fv[dleft][T.gi (n2)];
	x2 = 3./4.;
	y2 = 1./4.;
      }
      else
	v2 = v0;
    }
    else
      v2 = A.fv[dleft][T.gi (n2)];
  }
  else
    v2 = v0;
  return ((v1 - v0)*(x*y2 - x2*y) + (v2 - v0)*(x1*y - x*y1))/(x1*y2 - x2*y1) + v0;
}

// gfs_face_upwinded_value, src/advection.c:267-343
__device__ inline double face_upwinded_value (const Topo & T, const UpwindArgs & A, const Face & face,
					      int centered)
{
  double un;
  if (centered) {
    DevReader R = { A.u[face.d/2] };
    un = face_interpolated_value (T, face, R);
  }
  else
    un = A.un[face.d][T.gi (face.cell)];
  if (face.d & 1)
    un = - un;
  const double fc = A.fv[face.d][T.gi (face.cell)];
  if (!fine_coarse (face)) {
    const double fn = A.fv[face.d ^ 1][T.gi (face.neighbor)];
    return un > 0. ? fc : un < 0. ? fn : (fc + fn)/2.;
  }
  if (un > 0.)
    return fc;
  const int id = T.id (face.cell);
  const double vcoarse = T.dim == 2 ?
    adv_interpolate_1D1 (T, A, face.neighbor, face.d, perpendicular (face.d, id), 1./4.) :
    adv_interpolate_2D1 (T, A, face.neighbor, face.d, perpendicular3 (face.d, id, 0),
			 perpendicular3 (face.d, id, 1), 1./4., 1./4.);
  if (un == 0.)
    return (fc + vcoarse)/2.;
  return vcoarse;
}

// gfs_face_advected_normal_velocity, src/advection.c:513-539: the value of the face
__global__ void t_face_advected_un (Topo T, const FaceRec * faces, int n, UpwindArgs A, double * fval)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Face f = { faces[t].cell, faces[t].neighbor, faces[t].d };
  fval[t] = face_upwinded_value (T, A, f, 1);
}

// gfs_face_velocity_advection_flux, src/advection.c:398-435: the flux of the face
// gm == nullptr: gfs_face_advection_flux, src/advection.c:356-381 (a tracer)
__global__ void t_face_flux (Topo T, const FaceRec * faces, int n, UpwindArgs A, const double * gm,
			     double dt, double * fval)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Face f = { faces[t].cell, faces[t].neighbor, faces[t].d };
  DevReader G = { gm };
  double flux;
  if (gm) {
    flux = 1.*A.un[f.d][T.gi (f.cell)]*dt/T.size (f.cell);
    flux *= face_upwinded_value (T, A, f, 0) - face_interpolated_value (T, f, G)*dt/2.;
  }
  else
    flux = 1.*A.un[f.d][T.gi (f.cell)]*dt*face_upwinded_value (T, A, f, 0)/T.size (f.cell);
  if (f.d & 1)
    flux = - flux;
  fval[t] = flux;
}

// the flux sums of a leaf in traversal order, gfs_advection_update (src/advection.c:784-819) and
// add_pressure_gradient (src/timestep.c:809-812)
__global__ void t_gather_flux (Topo T, const Cell * cells, int n, const FaceRec * faces,
			       const int * inc_off, const int * inc, const double * fval,
			       double * v, const double * g, double dt, double gsrc = 0.)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  double f = 0.;
  for (int k = inc_off[t]; k < inc_off[t + 1]; k++) {
    const int fi = inc[k] >> 1, role = inc[k] & 1;
    const FaceRec & F = faces[fi];
    const double flux = fval[fi];
    if (role == 0)
      f -= flux;
    else if (F.neighbor.l == F.cell.l)
      f += flux;
    else
      f += flux/T.nc ();
  }
  const int gi = T.gi (cells[t]);
  double x = v[gi];
  x += f/1.;
  if (g)
    x -= g[gi]*dt;
  if (gsrc != 0.) {     /* gfs_domain_variable_centered_sources, src/source.c:62-108: a GfsSource */
    double sum = 0;
    sum += gsrc;
    x += dt*sum;
  }
  v[gi] = x;
}

// gfs_divergence, src/fluid.c:2357-2376: the derived variable `Divergence' of the leaves
__global__ void t_divergence_centered (Topo T, const Cell * cells, int n, P3 u, double * div)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Cell cell = cells[t];
  double d = 0.;
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < T.nd (); f.d++) {
    f.neighbor = T.neighbor (cell, f.d);
    if (exists (f.neighbor)) {
      DevReader R = { u.p[f.d/2] };
      d += 1.*((f.d & 1) ? -1. : 1.)*face_interpolated_value_generic (T, f, R);
    }
  }
  div[T.gi (cell)] = d/(1.*T.size (cell));
}

// gfs_domain_cfl, src/domain.c:2824-2923: the minimum of (length/|u|)^2 over faces and cells
__global__ void t_cfl_faces (Topo T, const FaceRec * faces, int n, P6 un, double * red)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  double best = DBL_MAX;
  if (t < n) {
    const double u = un.p[faces[t].d][T.gi (faces[t].cell)];
    if (u != 0.) {
      const double cflu = T.size (faces[t].cell)/fabs (u);
      best = cflu*cflu;
    }
  }
  best = wave_min (best);
  if ((threadIdx.x & 63) == 0 && best != DBL_MAX)
    atomic_min_pos (red, best);
}

struct D3 { double d[3]; };
__global__ void t_cfl_cells (Topo T, const Cell * cells, int n, P3 u, double * red, D3 visc, D3 src)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  double best = DBL_MAX;
  const bool in = t < n;
  if (!in) t = 0;
  const int g = T.gi (cells[t]);
  const double length = T.size (cells[t]);
  for (int c = 0; c < T.dim; c++) {
    const double fm = 1.;
    if (in && u.p[c][g] != 0.) {
      const double cflu = length/fabs (fm*u.p[c][g]);
      best = fmin (best, cflu*cflu);
    }
    if (visc.d[c] != 0. || src.d[c] != 0.) {      /* p->v[c]->sources: the acceleration scale, src/domain.c:2882-2891 */
      DevReader R = { u.p[c] };
      double gs = 0.;
      if (visc.d[c] != 0.)
	gs += source_diffusion_value (T, cells[t], R, visc.d[c]);
      if (src.d[c] != 0.)
	gs += src.d[c];
      if (in && gs != 0.) {
	const double cflg = 2.*length/fabs (fm*gs);
	best = fmin (best, cflg);
      }
    }
  }
  best = wave_min (best);
  if ((threadIdx.x & 63) == 0 && best != DBL_MAX)
    atomic_min_pos (red, best);
}

inline int blocks (int n) { return (n + 255)/256; }

// ---- host: tree construction and lists --------------------------------------------------------

struct Builder {
  int dim = 2;
  std::vector<std::vector<unsigned char>> flag;     // per level, (n + 2)^dim
  int r (int l) const { return (1 << l) + 2; }
  size_t idx (int l, int i, int j, int k) const { return i + (size_t) r (l)*(j + (dim == 3 ? (size_t) r (l)*k : 0)); }
  // cells of the dense levels allocated so far; the limit of gfship_tree_create (32-bit cell indices
  // times the number of directions in the tables) is enforced BEFORE a level is allocated: a Refine
  // function asking for level 10 in 3-D would otherwise allocate 1, 8.6, 68 GB on the host first
  long long total = 0;
  bool too_large = false;
  static constexpr long long MAX_CELLS = 1ll << 27;
  bool ensure (int l) {
    if ((int) flag.size () <= l) flag.resize (l + 1);
    if (flag[l].empty ()) {
      const long long cells = (long long) r (l)*r (l)*(dim == 3 ? r (l) : 1);
      if (total + cells > MAX_CELLS) { too_large = true; return false; }
      total += cells;
      flag[l].assign ((size_t) cells, NONE);
    }
    return true;
  }
  // oct_new with check_neighbors, src/ftt.c:45-83
  int refine_single (int l, int i, int j, int k) {
    if (l + 1 > GFSHIP_MAXLEVEL) return GFSHIP_EUNSUPPORTED;
    static const int di[6] = { 1, -1, 0, 0, 0, 0 }, dj[6] = { 0, 0, 1, -1, 0, 0 }, dk[6] = { 0, 0, 0, 0, 1, -1 };
    const int n = 1 << l;
    for (int d = 0; d < 2*dim; d++) {
      const int ni = i + di[d], nj = j + dj[d], nk = k + dk[d];
      if (ni < 1 || nj < 1 || ni > n || nj > n || (dim == 3 && (nk < 1 || nk > n)))
	continue;                   // the ghost trees are matched at the end (gfs_domain_match)
      if (flag[l][idx (l, ni, nj, nk)] == NONE) {
	const int pi = (ni + 1)/2, pj = (nj + 1)/2, pk = (nk + 1)/2;
	if (flag[l - 1][idx (l - 1, pi, pj, pk)] == LEAF) {
	  int e = refine_single (l - 1, pi, pj, pk);
	  if (e) return e;
	}
      }
    }
    if (!ensure (l + 1)) return GFSHIP_EUNSUPPORTED;
    flag[l][idx (l, i, j, k)] = NODE;
    for (int c = 0; c < (1 << dim); c++)
      flag[l + 1][idx (l + 1, 2*i - 1 + (c & 1), 2*j - ((c >> 1) & 1), 2*k - ((c >> 2) & 1))] = LEAF;
    return 0;
  }
  // ftt_cell_refine (src/ftt.c:169-192) with refine_maxlevel (src/refine.c:35-38)
  int refine_rec (int l, int i, int j, int k, gfship_refine_fn fn, void * ctx) {
    if (flag[l][idx (l, i, j, k)] == LEAF) {
      const double h = 1./(1 << l);
      const double x = -0.5 + (i - 0.5)*h, y = -0.5 + (j - 0.5)*h, z = dim == 3 ? -0.5 + (k - 0.5)*h : 0.;
      if (!(l < (* fn) (x, y, z, ctx)))
	return 0;
      int e = refine_single (l, i, j, k);
      if (e) return e;
    }
    for (int c = 0; c < (1 << dim); c++) {
      int e = refine_rec (l + 1, 2*i - 1 + (c & 1), 2*j - ((c >> 1) & 1), 2*k - ((c >> 2) & 1), fn, ctx);
      if (e) return e;
    }
    return 0;
  }
};

typedef std::function<void (Cell)> CellFn;
enum { T_ALL, T_LEAFS, T_NON_LEAFS, T_LEVEL, T_LEVEL_LEAFS, T_LEVEL_NON_LEAFS };

// ftt_cell_traverse, src/ftt.c:689-926 (pre-order; the post-order loops are done level by level)
void traverse (const Topo & T, Cell c, int flags, int max_depth, const CellFn & fn)
{
  const bool leaf = T.leaf (c);
  bool visit = false, descend = !leaf;
  if (flags == T_ALL || flags == T_LEAFS || flags == T_NON_LEAFS) {
    if (max_depth >= 0 && c.l > max_depth)
      return;
    visit = flags == T_ALL || (flags == T_LEAFS ? leaf : !leaf);
  }
  else if (flags == T_LEVEL) {
    visit = c.l == max_depth;
    descend = !visit && !leaf;
  }
  else if (flags == T_LEVEL_LEAFS) {
    visit = c.l == max_depth || leaf;
    descend = !visit;
  }
  else {
    visit = c.l == max_depth && !leaf;
    descend = !visit && !leaf;
  }
  if (visit)
    fn (c);
  if (descend)
    for (int k = 0; k < T.nc (); k++) {
      const Cell ch = T.child (c, k);
      if (exists (ch))
	traverse (T, ch, flags, max_depth, fn);
    }
}

inline Cell root_cell (const Topo & T) { return T.make (0, 1, 1, 1); }

// ftt_refine_corner, src/ftt.c:2013-2074
bool refine_corner (const Topo & T, Cell cell)
{
  static const int perp2[4][2] = { {2, 3}, {2, 3}, {1, 0}, {1, 0} };
  static const int perp3[6][4][2] =
    {{{4,2},{4,3},{5,2},{5,3}}, {{4,2},{4,3},{5,2},{5,3}},
     {{4,1},{4,0},{5,1},{5,0}}, {{4,1},{4,0},{5,1},{5,0}},
     {{2,1},{2,0},{3,1},{3,0}}, {{2,1},{2,0},{3,1},{3,0}}};
  for (int i = 0; i < T.nd (); i++) {
    const Cell nb = T.neighbor (cell, i);
    if (exists (nb) && !T.leaf (nb))
      for (int j = 0; j < T.ncd (); j++) {
	const Cell c = T.child_direction (nb, i ^ 1, j);
	if (exists (c)) {
	  if (T.dim == 2) {
	    const Cell nc = T.neighbor (c, perp2[i][j]);
	    if (exists (nc) && !T.leaf (nc))
	      return true;
	  }
	  else
	    for (int w = 0; w < 2; w++) {
	      const Cell nc = T.neighbor (c, perp3[i][j][w]);
	      if (exists (nc) && !T.leaf (nc))
		return true;
	    }
	  if (!T.leaf (c))
	    return true;
	}
      }
  }
  return false;
}

// traverse_face, src/ftt_internal.c:1-42 (leaves, max_depth = -1; the FTT_FLAG_TRAVERSED check
// only matters for the second pass, whose neighbours are ghost cells)
void traverse_face (const Topo & T, Cell cell, int d, std::vector<FaceRec> & out)
{
  FaceRec f = { cell, T.neighbor (cell, d), d };
  if (!exists (f.neighbor))
    return;
  if (T.leaf (cell) && !T.leaf (f.neighbor)) {
    const Cell coarse = cell, node = f.neighbor;
    f.d = d ^ 1;
    f.neighbor = coarse;
    for (int i = 0; i < T.ncd (); i++) {
      f.cell = T.child_direction (node, f.d, i);
      if (exists (f.cell))
	out.push_back (f);
    }
  }
  else
    out.push_back (f);
}

bool touches_side (const Topo & T, Cell c, int d)
{
  const int i = T.ci (c), j = T.cj (c), k = T.ck (c), n = T.n (c.l);
  return (d == 0 && i == n) || (d == 1 && i == 1) || (d == 2 && j == n) || (d == 3 && j == 1) ||
    (d == 4 && k == n) || (d == 5 && k == 1);
}

// ftt_face_traverse (src/ftt.c:2152-2215) as called by gfs_domain_face_traverse: kind 0 = FTT_XYZ,
// 1 + c = component c
void face_list (const Topo & T, const std::vector<Cell> & leaves, int kind, std::vector<FaceRec> & out)
{
  if (kind == 0) {
    for (Cell c : leaves)
      for (int d = 0; d < T.nd (); d += 2)
	traverse_face (T, c, d, out);
    for (int d = 1; d < T.nd (); d += 2)
      for (Cell c : leaves)
	if (touches_side (T, c, d))
	  traverse_face (T, c, d, out);
  }
  else {
    const int c0 = kind - 1;
    for (Cell c : leaves)
      traverse_face (T, c, 2*c0, out);
    for (Cell c : leaves)
      if (touches_side (T, c, 2*c0 + 1))
	traverse_face (T, c, 2*c0 + 1, out);
  }
}

// gfship_tree_host_check builds the host side of a tree (flags, tables, lists, dependency levels,
// compiled stencils, loop plans) without a device: nothing is uploaded
bool g_host_only = false;

template <class X> int to_device (const std::vector<X> & h, X ** d)
{
  *d = nullptr;
  if (h.empty () || g_host_only) return 0;
  GFSHIP_HIP (hipMalloc ((void **) d, h.size ()*sizeof (X)));
  GFSHIP_HIP (hipMemcpy (*d, h.data (), h.size ()*sizeof (X), hipMemcpyHostToDevice));
  return 0;
}

// the ghost cells of a selection of the traversal and their periodic images
void ghost_list (const Topo & T, const int * sides, int flags, int max_depth, std::vector<Ghost> & out)
{
  for (int l = 0; l <= T.depth; l++) {
    if (max_depth >= 0 && l > max_depth)
      break;
    const int n = T.n (l), r = T.r (l);
    for (int side = 0; side < T.nd (); side++)
      for (int tb = 1; tb <= (T.dim == 3 ? n : 1); tb++)
	for (int ta = 1; ta <= n; ta++) {
	  int g[3], im[3];
	  const int a = side/2, o1 = a == 0 ? 1 : 0, o2 = a == 2 ? 1 : 2;
	  g[a] = (side & 1) ? 0 : n + 1;
	  im[a] = (side & 1) ? n : 1;
	  if (sides[side] != GFSHIP_SIDE_PERIODIC)     /* GfsBoundary: the cell the ghost touches */
	    im[a] = (side & 1) ? 1 : n;
	  g[o1] = im[o1] = ta;
	  g[o2] = im[o2] = T.dim == 3 ? tb : 0;
	  const int G = g[0] + r*(g[1] + r*g[2]), I = im[0] + r*(im[1] + r*im[2]);
	  const unsigned char f = T.flag[T.off[l] + G];
	  if (f == NONE)
	    continue;
	  const bool take = flags == T_LEAFS ? f == LEAF : (l == max_depth || f == LEAF);
	  if (take) {
	    Ghost gh = { T.off[l] + G, T.off[l] + I, side, l };
	    out.push_back (gh);
	  }
	}
  }
}

int stencil_tape (gfship_tree * tr, int m, Sweep * S, const std::vector<Cell> & sorted,
		  const std::vector<int> & lev_off);
Sgn6 homogeneous_signs (const gfship_tree * tr);

// dependency levels of an exact-order sweep over `order': a cell runs after every earlier cell
// it reads (it must see the new value) and after every earlier cell that reads it (which must
// still see the old one)
int sweep_plan (gfship_tree * tr, int m, Sweep * S)
{
  const Topo & T = tr->H;
  std::vector<Cell> order;
  traverse (T, root_cell (T), T_LEVEL_LEAFS, m, [&] (Cell c) { order.push_back (c); });
  std::vector<int> pos (tr->ncell, -1);
  for (size_t k = 0; k < order.size (); k++)
    pos[T.gi (order[k])] = (int) k;
  std::vector<int> lev (order.size (), 0), minlev (order.size (), 0), reads;
  Recorder R = { &reads };
  int nlev = 0;
  for (size_t k = 0; k < order.size (); k++) {
    reads.clear ();
    relax_cell (T, order[k], R, 0., 1., m);
    int L = minlev[k];
    for (int g : reads) {
      const int p = pos[g];
      if (p >= 0 && p < (int) k)
	L = std::max (L, lev[p] + 1);
    }
    lev[k] = L;
    for (int g : reads) {
      const int p = pos[g];
      if (p > (int) k)
	minlev[p] = std::max (minlev[p], L + 1);
    }
    nlev = std::max (nlev, L + 1);
  }
  if (getenv ("GFSHIP_TREE_DEBUG")) {
    // what pipelining the nrelax sweeps of a loop would give: the sequential program [copies of the
    // ghosts][sweep 0][copies][sweep 1]... scheduled into levels that keep its RAW / WAR / WAW order
    const int nrelax = 4;
    std::vector<Ghost> gh0;
    ghost_list (T, tr->side, T_LEVEL_LEAFS, m, gh0);
    std::vector<int> wlev (tr->ncell, 0), rlev (tr->ncell, 0);
    int tot = 0;
    std::vector<std::vector<int>> rd (order.size ());
    for (size_t k = 0; k < order.size (); k++) {
      reads.clear ();
      relax_cell (T, order[k], R, 0., 1., m);
      rd[k] = reads;
    }
    for (int sw = 0; sw < nrelax; sw++) {
      for (const Ghost & G : gh0) {
	int L = std::max (wlev[G.img], std::max (rlev[G.g], wlev[G.g])) + 1;
	rlev[G.img] = std::max (rlev[G.img], L);
	wlev[G.g] = L; rlev[G.g] = 0;
	tot = std::max (tot, L);
      }
      for (size_t k = 0; k < order.size (); k++) {
	const int w = T.gi (order[k]);
	int L = std::max (rlev[w], wlev[w]);
	for (int g : rd[k]) L = std::max (L, wlev[g]);
	L++;
	for (int g : rd[k]) rlev[g] = std::max (rlev[g], L);
	wlev[w] = L; rlev[w] = 0;
	tot = std::max (tot, L);
      }
    }
    fprintf (stderr, "gfship_tree: sweep of level %d: %zu cells, %d dependency levels per sweep, %d x %d = %d sequential, %d pipelined\n",
	     m, order.size (), nlev, nrelax, nlev, nrelax*nlev, tot);
  }
  std::vector<int> off (nlev + 1, 0);
  for (size_t k = 0; k < order.size (); k++)
    off[lev[k] + 1]++;
  for (int L = 0; L < nlev; L++)
    off[L + 1] += off[L];
  std::vector<Cell> sorted (order.size ());
  std::vector<int> cur (off.begin (), off.end () - 1);
  for (size_t k = 0; k < order.size (); k++)
    sorted[cur[lev[k]]++] = order[k];
  std::vector<Ghost> gh;
  ghost_list (T, tr->side, T_LEVEL_LEAFS, m, gh);
  S->ncells = (int) order.size ();
  S->nlev = nlev;
  S->nghosts = (int) gh.size ();
  S->h_ghosts = gh;
  int e;
  if ((e = to_device (sorted, &S->cells)) || (e = to_device (off, &S->lev_off)) ||
      (e = to_device (gh, &S->ghosts)))
    return e;
  return stencil_tape (tr, m, S, sorted, off);
}

// ---- compiled stencils: the host side ---------------------------------------------------------

struct TapeOut { std::vector<int> ti, tv; std::vector<double> td; };

// interpolate_1D1 / interpolate_2D1 in the coarse cell A towards the fine cell (face: fine -> A in
// direction dface), src/fluid.c:178-245: the terms of p.b go to the streams, p.a is returned
double tape_interpolation_gen (const Topo & T, Cell fine, Cell A, int dface, TapeOut & o)
{
  const int id = T.id (fine);
  int dirs[2], ndirs;
  if (T.dim == 2) { dirs[0] = perpendicular (dface, id); ndirs = 1; }
  else { dirs[0] = perpendicular3 (dface, id, 0); dirs[1] = perpendicular3 (dface, id, 1); ndirs = 2; }
  double pa = 1.;
  const size_t nt_at = o.ti.size ();
  o.ti.push_back (0);
  int nt = 0;
  for (int w = 0; w < ndirs; w++) {
    const Cell nb = T.neighbor (A, dirs[w]);
    if (!exists (nb))
      continue;
    double x2 = 1.;
    std::vector<int> idx;
    int cnt = 0;
    if (T.leaf (nb))
      idx.push_back (T.gi (nb));
    else {
      for (int i = 0; i < T.ncd (); i++) {
	const Cell ch = T.child_direction (nb, dirs[w] ^ 1, i);
	if (exists (ch)) { idx.push_back (T.gi (ch)); cnt++; }
      }
      if (cnt > 0)
	x2 = 3./4.;
      else
	idx.push_back (T.gi (A));       /* average_neighbor_value falls back on the cell of the face */
    }
    const double a = (1./4.)/x2;
    o.td.push_back (a);
    o.ti.push_back (cnt);
    for (int g : idx) o.tv.push_back (g);
    pa -= a;
    nt++;
  }
  o.ti[nt_at] = nt;
  return pa;
}

// one cell of the sweep of level max_level: what face_gradient (tree.hpp) does for each direction
void tape_cell_gen (const Topo & T, Cell cell, int max_level, TapeOut & o)
{
  o.tv.push_back (T.gi (cell));          /* the value of the cell itself */
  for (int d = 0; d < T.nd (); d++) {
    const Cell nb = T.neighbor (cell, d);
    if (!exists (nb)) { o.ti.push_back (K_NONE); continue; }
    if (nb.l < cell.l) {
      o.ti.push_back (K_FC);
      const size_t cb_at = o.td.size ();
      o.td.push_back (0.);
      o.tv.push_back (T.gi (nb));
      const double pa = tape_interpolation_gen (T, cell, nb, d, o);
      o.td[cb_at] = 2.*pa/3.;
    }
    else if (cell.l == max_level || T.leaf (nb)) {
      o.ti.push_back (K_SAME);
      o.tv.push_back (T.gi (nb));
    }
    else {
      o.ti.push_back (K_DEEP);
      const size_t n_at = o.ti.size ();
      o.ti.push_back (0);
      int nch = 0;
      for (int i = 0; i < T.ncd (); i++) {
	const Cell ch = T.child_direction (nb, d ^ 1, i);
	if (!exists (ch))
	  continue;
	const size_t gb_at = o.td.size ();
	o.td.push_back (0.);
	o.tv.push_back (T.gi (ch));
	const double pa = tape_interpolation_gen (T, ch, cell, d ^ 1, o);
	o.td[gb_at] = 2.*pa/3.;
	nch++;
      }
      o.ti[n_at] = nch;
    }
  }
}

int stencil_tape (gfship_tree * tr, int m, Sweep * S, const std::vector<Cell> & sorted,
		  const std::vector<int> & lev_off)
{
  const Topo & T = tr->H;
  TapeOut o;
  std::vector<int> cell_off;
  for (size_t c = 0; c < sorted.size (); c++) {
    cell_off.push_back ((int) o.ti.size ()); cell_off.push_back ((int) o.td.size ()); cell_off.push_back ((int) o.tv.size ());
    tape_cell_gen (T, sorted[c], m, o);
  }
  cell_off.push_back ((int) o.ti.size ()); cell_off.push_back ((int) o.td.size ()); cell_off.push_back ((int) o.tv.size ());
  // chunks: consecutive cells of one dependency level whose streams fit the LDS
  std::vector<int> chunk;
  for (size_t L = 0; L + 1 < lev_off.size (); L++) {
    int c = lev_off[L];
    while (c < lev_off[L + 1]) {
      chunk.push_back (c);
      int e = c;
      while (e < lev_off[L + 1]) {
	const size_t bytes = 8*(size_t) (cell_off[3*(e + 1) + 2] - cell_off[3*c + 2]) +
	  8*(size_t) (cell_off[3*(e + 1) + 1] - cell_off[3*c + 1]) + 4*(size_t) (cell_off[3*(e + 1)] - cell_off[3*c]) + 8;
	if (bytes > TAPE_LDS_BYTES)
	  break;
	e++;
      }
      if (e == c) return GFSHIP_EUNSUPPORTED;      /* one cell larger than the LDS: cannot happen */
      c = e;
    }
  }
  S->nchunks = (int) chunk.size ();
  chunk.push_back ((int) sorted.size ());
  int e;
  if ((e = to_device (o.ti, &S->ti)) || (e = to_device (o.td, &S->td)) || (e = to_device (o.tv, &S->tv)) ||
      (e = to_device (cell_off, &S->cell_off)) || (e = to_device (chunk, &S->chunk)))
    return e;
  if (o.td.empty ())      /* to_device leaves nullptr for an empty stream: never dereferenced */
    S->td = nullptr;
  S->taped = true;
  S->h_ti = o.ti; S->h_tv = o.tv; S->h_td = o.td; S->h_cell_off = cell_off;
  S->h_g.resize (sorted.size ());
  for (size_t c = 0; c < sorted.size (); c++) S->h_g[c] = T.gi (sorted[c]);
  return 0;
}

void loop_free (Sweep::Loop & P)
{
  (void) hipFree (P.ti); (void) hipFree (P.tv); (void) hipFree (P.td);
  (void) hipFree (P.node_off); (void) hipFree (P.node_g); (void) hipFree (P.chunk);
  (void) hipFree (P.desc); (void) hipFree (P.rec);
  if (P.flow) { flow_free (*P.flow); delete P.flow; }
  P = Sweep::Loop ();
}

// The relax loop of level m as one program: [copies of the ghosts][sweep 0 in tree order][copies]
// [sweep 1] ... [sweep nrelax - 1], scheduled into levels that keep the order of every read after the
// write it must see, and of every write after the reads of the value it replaces (RAW, WAR, WAW of
// the sequential program): the results are those of the sequential program, and sweep s + 1 follows
// sweep s a few levels behind instead of waiting for its end
int loop_plan (gfship_tree * tr, int m, unsigned nrelax, Sweep * S, Sweep::Loop * out = nullptr,
	       const Sgn6 * signs = nullptr)
{
  Sweep::Loop & P = out ? *out : S->loop;
  loop_free (P);
  const Sgn6 sg = signs ? *signs : homogeneous_signs (tr);
  const size_t nc = S->h_g.size ();
  // the cells were sorted by single-sweep dependency level; the sequential order inside a sweep is
  // the tree order: recover it from the positions the sweep plan kept (sorted is stable per level,
  // and any order consistent with the single-sweep levels is a valid sequential order)
  struct Node { int kind, idx, sweep, level; };   /* kind 0: cell idx of the sweep, 1: ghost idx */
  std::vector<Node> nodes;
  nodes.reserve (nrelax*(nc + S->h_ghosts.size ()));
  std::vector<int> wlev (tr->ncell, 0), rlev (tr->ncell, 0);
  int nlev = 0;
  for (unsigned sw = 0; sw < nrelax; sw++) {
    for (size_t q = 0; q < S->h_ghosts.size (); q++) {
      const Ghost & G = S->h_ghosts[q];
      const int L = std::max (wlev[G.img], std::max (rlev[G.g], wlev[G.g])) + 1;
      rlev[G.img] = std::max (rlev[G.img], L);
      wlev[G.g] = L; rlev[G.g] = 0;
      nodes.push_back ({ 1, (int) q, (int) sw, L });
      nlev = std::max (nlev, L);
    }
    for (size_t c = 0; c < nc; c++) {
      const int w = S->h_g[c];
      int L = std::max (rlev[w], wlev[w]);
      for (int k = S->h_cell_off[3*c + 2]; k < S->h_cell_off[3*(c + 1) + 2]; k++)
	L = std::max (L, wlev[S->h_tv[k]]);
      L++;
      for (int k = S->h_cell_off[3*c + 2]; k < S->h_cell_off[3*(c + 1) + 2]; k++)
	rlev[S->h_tv[k]] = std::max (rlev[S->h_tv[k]], L);
      wlev[w] = L; rlev[w] = 0;
      nodes.push_back ({ 0, (int) c, (int) sw, L });
      nlev = std::max (nlev, L);
    }
  }
  std::stable_sort (nodes.begin (), nodes.end (), [] (const Node & a, const Node & b) { return a.level < b.level; });
  std::vector<int> ti, tv, node_off, node_g, chunk;
  std::vector<double> td;
  for (const Node & N : nodes) {
    node_off.push_back ((int) ti.size ()); node_off.push_back ((int) td.size ()); node_off.push_back ((int) tv.size ());
    if (N.kind == 1) {
      const Ghost & G = S->h_ghosts[N.idx];
      ti.push_back (K_GHOST);
      td.push_back (sg.s[G.side]);
      tv.push_back (G.img);
      node_g.push_back (G.g);
    }
    else {
      const int c = N.idx;
      ti.insert (ti.end (), S->h_ti.begin () + S->h_cell_off[3*c], S->h_ti.begin () + S->h_cell_off[3*(c + 1)]);
      td.insert (td.end (), S->h_td.begin () + S->h_cell_off[3*c + 1], S->h_td.begin () + S->h_cell_off[3*(c + 1) + 1]);
      tv.insert (tv.end (), S->h_tv.begin () + S->h_cell_off[3*c + 2], S->h_tv.begin () + S->h_cell_off[3*(c + 1) + 2]);
      node_g.push_back (S->h_g[c]);
    }
  }
  node_off.push_back ((int) ti.size ()); node_off.push_back ((int) td.size ()); node_off.push_back ((int) tv.size ());
  // chunks: consecutive nodes of one level whose streams fit the LDS
  size_t c = 0;
  while (c < nodes.size ()) {
    chunk.push_back ((int) c);
    size_t e = c;
    while (e < nodes.size () && nodes[e].level == nodes[c].level) {
      const size_t bytes = 8*(size_t) (node_off[3*(e + 1) + 2] - node_off[3*c + 2]) +
	8*(size_t) (node_off[3*(e + 1) + 1] - node_off[3*c + 1]) + 4*(size_t) (node_off[3*(e + 1)] - node_off[3*c]) + 8;
      if (bytes > TAPE_LDS_BYTES)
	break;
      e++;
    }
    if (e == c) return GFSHIP_EUNSUPPORTED;
    c = e;
  }
  for (int d = 0; d < 6; d++) P.sg[d] = sg.s[d];
  P.nchunks = (int) chunk.size ();
  chunk.push_back ((int) nodes.size ());
  P.nnodes = (int) nodes.size ();
  P.nlev = nlev;
  int e;
  if ((e = to_device (ti, &P.ti)) || (e = to_device (td, &P.td)) || (e = to_device (tv, &P.tv)) ||
      (e = to_device (node_off, &P.node_off)) || (e = to_device (node_g, &P.node_g)) ||
      (e = to_device (chunk, &P.chunk)))
    return e;
  {
    std::vector<int> desc, rec;
    for (int k = 0; k < P.nchunks; k++) {
      const int c0 = chunk[k], c1 = chunk[k + 1];
      const int v[8] = { c0, c1, node_off[3*c0], node_off[3*c1], node_off[3*c0 + 1], node_off[3*c1 + 1],
			 node_off[3*c0 + 2], node_off[3*c1 + 2] };
      desc.insert (desc.end (), v, v + 8);
    }
    for (int q = 0; q < 8; q++) desc.push_back (0);          /* the descriptor read past the last chunk */
    for (size_t c = 0; c < nodes.size (); c++) {
      rec.push_back (node_off[3*c]); rec.push_back (node_off[3*c + 1]); rec.push_back (node_off[3*c + 2]);
      rec.push_back (node_g[c]);
    }
    for (int q = 0; q < 4; q++) rec.push_back (0);
    if ((e = to_device (desc, &P.desc)) || (e = to_device (rec, &P.rec)))
      return e;
  }
  P.nrelax = nrelax;
  if (g_host_only) {
    P.h_ti = ti; P.h_tv = tv; P.h_td = td; P.h_node_off = node_off; P.h_node_g = node_g;
    for (const Node & N : nodes) P.h_node_level.push_back (N.level);
  }
  {
    // the same loop as a flow plan (tree_flow.hpp)
    std::vector<int> level_of (tr->ncell, 0);
    for (int l = 0; l <= tr->H.depth; l++)
      for (int g = tr->H.off[l]; g < tr->H.off[l + 1]; g++) level_of[g] = l;
    P.flow = new FlowPlan;
    static int flow_width = -1;
    if (flow_width < 0) {
      const char * e = getenv ("GFSHIP_FLOW_WIDTH");      /* lab: operations per level (a multiple of 64 up to 512) */
      flow_width = e ? std::min (FLOW_WIDTH, std::max (64, atoi (e)/64*64)) : 0;
    }
    /* a level costs about the same whatever its width: the widest plan wins on octrees; the levels of a quadtree
       are narrow, 256 operations per level are as few levels and fewer idle wavefronts (measured: 11.5 against
       12.5 ms per step on the quadtree bench) */
    const int width = flow_width ? flow_width : tr->H.dim == 2 ? 256 : FLOW_WIDTH;
    if (!flow_plan (tr->ncell, tr->H.dim, S, nrelax, sg, level_of.data (), tr->H.dim == 2, P.flow, g_host_only, width)) {
      flow_free (*P.flow);
      delete P.flow;
      P.flow = nullptr;
    }
  }
  if (getenv ("GFSHIP_TREE_DEBUG"))
    fprintf (stderr, "gfship_tree: relax loop of level %d: %u sweeps, %d nodes in %d levels, %d chunks; flow plan: %d operations in %d levels, %d constants\n",
	     m, nrelax, P.nnodes, P.nlev, P.nchunks, P.flow ? P.flow->nops : -1, P.flow ? P.flow->nlev : -1,
	     P.flow ? P.flow->nct : -1);
  return 0;
}

int face_set (gfship_tree * tr, int kind, FaceSet * F)
{
  const Topo & T = tr->H;
  std::vector<FaceRec> faces;
  face_list (T, tr->hleaves, kind, faces);
  std::vector<int> leafpos (tr->ncell, -1);
  for (size_t k = 0; k < tr->hleaves.size (); k++)
    leafpos[T.gi (tr->hleaves[k])] = (int) k;
  std::vector<std::vector<int>> inc (tr->hleaves.size ());
  for (size_t k = 0; k < faces.size (); k++) {
    const int a = leafpos[T.gi (faces[k].cell)], b = leafpos[T.gi (faces[k].neighbor)];
    if (a >= 0) inc[a].push_back ((int) (k << 1));
    if (b >= 0) inc[b].push_back ((int) (k << 1 | 1));
  }
  std::vector<int> off (inc.size () + 1, 0), flat;
  for (size_t k = 0; k < inc.size (); k++) {
    off[k + 1] = off[k] + (int) inc[k].size ();
    flat.insert (flat.end (), inc[k].begin (), inc[k].end ());
  }
  F->nfaces = (int) faces.size ();
  int e;
  if ((e = to_device (faces, &F->faces)) || (e = to_device (off, &F->inc_off)) ||
      (e = to_device (flat, &F->inc)))
    return e;
  if (!g_host_only)
    GFSHIP_HIP (hipMalloc ((void **) &F->fval, std::max<size_t> (1, faces.size ())*sizeof (double)));
  return 0;
}

#define KCHECK() GFSHIP_HIP (hipGetLastError ())

// ---- host: the algorithms ----------------------------------------------------------------------

// the homogeneous form of the conditions (gfs_domain_homogeneous_bc): ghost = s * cell
Sgn6 homogeneous_signs (const gfship_tree * tr)
{
  Sgn6 sg;
  for (int d = 0; d < 6; d++)
    sg.s[d] = tr->side[d] != GFSHIP_SIDE_PERIODIC && tr->bc_p[d] == GFSHIP_BC_DIRICHLET ? -1. : 1.;
  return sg;
}

// gfs_domain_bc of P on the leaves of a tree with GfsBoundary sides
int bc_solution (gfship_tree * tr, double * v)
{
  if (!tr->nghost_leaves) return 0;
  Sgn6 kind;
  for (int d = 0; d < 6; d++)
    kind.s[d] = tr->side[d] == GFSHIP_SIDE_PERIODIC ? -1. : (double) tr->bc_p[d];
  t_bc_values<<<blocks (tr->nghost_leaves), 256, 0, tr->stream>>> (tr->ghost_leaves, tr->nghost_leaves, v,
      tr->var[V_BCVAL], kind);
  GFSHIP_HIP (hipGetLastError ());
  return 0;
}

int bc_leaves (gfship_tree * tr, double * v, int comp = -1)
{
  if (tr->has_boundary && comp >= 0) {      /* a vector component: symmetry on the GfsBoundary sides */
    Sgn6 sg;
    for (int d = 0; d < 6; d++)
      sg.s[d] = tr->side[d] != GFSHIP_SIDE_PERIODIC && comp == d/2 ? -1. : 1.;
    if (tr->nghost_leaves)
      t_copy_ghosts_signed<<<blocks (tr->nghost_leaves), 256, 0, tr->stream>>> (tr->ghost_leaves, tr->nghost_leaves, v, sg);
    GFSHIP_HIP (hipGetLastError ());
    return 0;
  }
  if (tr->has_boundary)      /* P and the other scalars: the conditions of P */
    return bc_solution (tr, v);
  if (tr->nghost_leaves)
    t_copy_ghosts<<<blocks (tr->nghost_leaves), 256, 0, tr->stream>>> (tr->ghost_leaves, tr->nghost_leaves, v);
  KCHECK ();
  return 0;
}

// gfs_domain_bc of velocity component c: the conditions of gfship_tree_set_bc_u (symmetry by default)
int bc_velocity (gfship_tree * tr, int c)
{
  bool plain = true;
  for (int d = 0; d < 6; d++)
    if (tr->side[d] != GFSHIP_SIDE_PERIODIC && tr->bc_u[c][d] != GFSHIP_BC_SYMMETRY) plain = false;
  if (plain || !tr->nghost_leaves)
    return bc_leaves (tr, tr->var[V_U + c], c);
  Sgn6 kind;
  for (int d = 0; d < 6; d++)
    kind.s[d] = tr->side[d] == GFSHIP_SIDE_PERIODIC ? -1. : (double) tr->bc_u[c][d];
  t_bc_values<<<blocks (tr->nghost_leaves), 256, 0, tr->stream>>> (tr->ghost_leaves, tr->nghost_leaves, tr->var[V_U + c],
      tr->var[V_BCU + c], kind, c);
  KCHECK ();
  return 0;
}

// the homogeneous conditions of velocity component c (gfs_domain_homogeneous_bc in the relax loops of its
// diffusion solve): ghost = s * cell
Sgn6 homogeneous_signs_u (const gfship_tree * tr, int c)
{
  Sgn6 sg;
  for (int d = 0; d < 6; d++) {
    sg.s[d] = 1.;
    if (tr->side[d] != GFSHIP_SIDE_PERIODIC) {
      const int k = tr->bc_u[c][d];
      sg.s[d] = k == GFSHIP_BC_DIRICHLET ? -1. : k == GFSHIP_BC_NEUMANN ? 1. : (c == d/2 ? -1. : 1.);
    }
  }
  return sg;
}

// post-order traversal of the non-leaf cells: deepest level first
// the same for several variables in one launch per level (gfs_cell_coarse_init restricts five to seven variables:
// thirty to forty launches of a few microseconds each per step otherwise)
struct PN { double * p[8]; };
__global__ void t_from_below_n (Topo T, const Cell * cells, int n, PN V, int nv)
{
  int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= n*nv) return;
  const Cell c = cells[t/nv];
  double * v = V.p[t % nv];
  double val = 0., sa = 0.;
  for (int k = 0; k < T.nc (); k++) {
    const Cell ch = T.child (c, k);
    if (exists (ch)) {
      val += v[T.gi (ch)]*1.;
      sa += 1.;
    }
  }
  v[T.gi (c)] = val/sa;
}

int from_below_n (gfship_tree * tr, const PN & V, int nv)
{
  for (int l = tr->H.depth - 1; l >= 0; l--)
    if (tr->nnonleaf[l]) {
      t_from_below_n<<<blocks (tr->nnonleaf[l]*nv), 256, 0, tr->stream>>> (tr->D, tr->nonleaf[l], tr->nnonleaf[l], V, nv);
      KCHECK ();
    }
  return 0;
}

int from_below (gfship_tree * tr, double * v, int mode)
{
  for (int l = tr->H.depth - 1; l >= 0; l--)
    if (tr->nnonleaf[l]) {
      t_from_below<<<blocks (tr->nnonleaf[l]), 256, 0, tr->stream>>> (tr->D, tr->nonleaf[l], tr->nnonleaf[l], v, mode);
      KCHECK ();
    }
  return 0;
}

int residual_norm (gfship_tree * tr, const double * u, const double * rhs, double * res, double dt,
		   gfship_norm * out)
{
  GFSHIP_HIP (hipMemsetAsync (tr->d_red, 0, 4*sizeof (double), tr->stream));
  {
    static int no_tape = -1;
    if (no_tape < 0) {
      const char * e = getenv ("GFSHIP_TREE_NO_RESIDUAL_TAPE");      /* 1: the residual by the code that walks the tree */
      no_tape = e ? atoi (e) : 0;
    }
    const Sweep & S = tr->sweep[tr->H.depth];
    if (!no_tape && S.taped && S.ncells == tr->nleaves)
      t_residual_tape<<<blocks (S.ncells), 256, 0, tr->stream>>> (tr->D, S.cells, S.cell_off, S.ncells, S.ti, S.td, S.tv,
								   u, rhs, res, tr->d_red);
    else
      t_residual<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, u, rhs, res, tr->d_red);
  }
  KCHECK ();
  GFSHIP_HIP (hipMemcpyAsync (tr->h_red, tr->d_red, 4*sizeof (double), hipMemcpyDeviceToHost, tr->stream));
  GFSHIP_HIP (hipStreamSynchronize (tr->stream));
  // gfs_norm_update + the scaling of gfs_domain_norm_residual, src/domain.c:2264-2288
  const double w = tr->nleaves;
  dt *= dt;
  out->bias = tr->h_red[1]*dt;
  out->first = tr->h_red[2]/w*dt;
  out->second = sqrt (tr->h_red[3]/w)*dt;
  out->infty = tr->h_red[0]*dt;
  out->w = w;
  return 0;
}

// op 0: the Poisson relax on V_DP with the conditions of P; op 1: diffusion_relax with the face weight w and
// the homogeneous conditions sg of the variable being diffused
int relax_loop (gfship_tree * tr, int m, unsigned nrelax, double omega, int op = 0, double w = 1.,
		const Sgn6 * sgp = nullptr)
{
  Sweep & S = tr->sweep[m];
  static int use_template = -1;
  if (use_template < 0) {
    const char * e = getenv ("GFSHIP_TREE_TEMPLATE_RELAX");   /* 1: the stencil code walks the tree in every sweep */
    use_template = e ? atoi (e) : 0;
  }
  if (!tr->tape_attr_set) {       /* more than 64 KB of dynamic LDS needs the attribute */
    GFSHIP_HIP (hipFuncSetAttribute ((const void *) t_relax_tape, hipFuncAttributeMaxDynamicSharedMemorySize,
				     TAPE_LDS_BYTES));
    tr->tape_attr_set = true;
  }
  static int no_pipeline = -1;
  if (no_pipeline < 0) {
    const char * e = getenv ("GFSHIP_TREE_NO_PIPELINE");      /* 1: sweep after sweep (t_relax_tape) */
    no_pipeline = e ? atoi (e) : 0;
  }
  const Sgn6 sg = sgp ? *sgp : homogeneous_signs (tr);
  if (S.taped && !use_template && (!no_pipeline || op == 1)) {
    Sweep::Loop * P = &S.loop;
    if (op == 1) {
      /* the plan with these conditions and this number of sweeps, kept from an earlier solve */
      P = nullptr;
      for (Sweep::Loop & Q : S.dloop)
	if (Q.nrelax == nrelax && Q.nnodes > 0 && !memcmp (Q.sg, sg.s, sizeof (Q.sg))) P = &Q;
      if (!P) {
	P = &S.dloop[S.dloop_next];
	S.dloop_next = (S.dloop_next + 1) % 6;
	int e = loop_plan (tr, m, nrelax, &S, P, &sg);
	if (e) return e;
      }
    }
    else if (S.loop.nrelax != nrelax || memcmp (S.loop.sg, sg.s, sizeof (S.loop.sg))) {
      int e = loop_plan (tr, m, nrelax, &S);
      if (e) return e;
    }
    static int no_flow = -1;
    if (no_flow < 0) {
      const char * e = getenv ("GFSHIP_TREE_NO_FLOW");          /* 1: the tape kernels (t_relax_nodes_pf) */
      no_flow = e ? atoi (e) : 0;
    }
    if (P->flow && !no_flow) {
      const FlowPlan & F = *P->flow;
      t_flow_pack<<<blocks (F.npos), 256, 0, tr->stream>>> (F.gidx, F.npos, tr->var[V_DP], tr->var[V_RES], F.up, F.rp);
      if (tr->H.dim == 3)
	t_relax_flow<3><<<1, F.width + 64, 0, tr->stream>>> (F.rec, F.lev_off, F.nlev, F.ct, F.nct, F.up, F.rp, F.npos, omega, op, w);
      else
	t_relax_flow<2><<<1, F.width + 64, 0, tr->stream>>> (F.rec, F.lev_off, F.nlev, F.ct, F.nct, F.up, F.rp, F.npos, omega, op, w);
      t_flow_unpack<<<blocks (F.npos), 256, 0, tr->stream>>> (F.gidx, F.npos, F.up, tr->var[V_DP]);
      KCHECK ();
      return 0;
    }
    static int old_nodes = -1;
    if (old_nodes < 0) {
      const char * e = getenv ("GFSHIP_TREE_NO_PREFETCH");      /* 1: t_relax_nodes (every load in place) */
      old_nodes = e ? atoi (e) : 0;
    }
    if (old_nodes) {
      GFSHIP_HIP (hipFuncSetAttribute ((const void *) t_relax_nodes, hipFuncAttributeMaxDynamicSharedMemorySize,
				       TAPE_LDS_BYTES));
      t_relax_nodes<<<1, 1024, TAPE_LDS_BYTES, tr->stream>>> (tr->D, P->node_g, P->node_off, P->chunk,
	  P->nchunks, P->ti, P->td, P->tv, tr->var[V_DP], tr->var[V_RES], omega, op, w);
    }
    else {
      GFSHIP_HIP (hipFuncSetAttribute ((const void *) t_relax_nodes_pf, hipFuncAttributeMaxDynamicSharedMemorySize,
				       TAPE_LDS_BYTES));
      t_relax_nodes_pf<<<1, 1024, TAPE_LDS_BYTES, tr->stream>>> (tr->D, P->rec, P->desc, P->nchunks, P->ti, P->td, P->tv,
	  tr->var[V_DP], tr->var[V_RES], omega, op, w, P->node_g, P->node_off);
    }
  }
  else if (S.taped && !use_template)
    t_relax_tape<<<1, 1024, TAPE_LDS_BYTES, tr->stream>>> (tr->D, S.cells, S.cell_off, S.ncells, S.chunk, S.nchunks,
	S.ti, S.td, S.tv, S.ghosts, S.nghosts, tr->var[V_DP], tr->var[V_RES], nrelax, omega, sg);
  else
    t_relax_loop<<<1, 1024, 0, tr->stream>>> (tr->D, S.cells, S.lev_off, S.nlev, S.ghosts, S.nghosts,
					       tr->var[V_DP], tr->var[V_RES], nrelax, omega, m, sg, op, w);
  KCHECK ();
  return 0;
}

// gfs_poisson_cycle, src/poisson.c:1105-1178 (dia = 0)
int poisson_cycle (gfship_tree * tr, gfship_multilevel_params * p, double * u, const double * rhs)
{
  int e;
  const unsigned minlevel = p->minlevel;
  if ((e = from_below (tr, tr->var[V_RES], 1))) return e;
  unsigned nrelax = p->nrelax;
  for (unsigned l = minlevel; l < p->depth; l++)
    nrelax *= p->erelax;
  // dp: the cells of the first level are reset, every other cell is written before it is read
  GFSHIP_HIP (hipMemsetAsync (tr->var[V_DP], 0, tr->ncell*sizeof (double), tr->stream));
  if ((e = relax_loop (tr, minlevel, nrelax, p->omega))) return e;
  nrelax /= p->erelax;
  for (unsigned m = minlevel + 1; m <= p->depth; m++, nrelax /= p->erelax) {
    if (tr->nnonleaf[m - 1]) {
      t_from_above<<<blocks (tr->nnonleaf[m - 1]), 256, 0, tr->stream>>> (tr->D, tr->nonleaf[m - 1], tr->nnonleaf[m - 1], tr->var[V_DP]);
      KCHECK ();
    }
    if ((e = relax_loop (tr, m, nrelax, p->omega))) return e;
  }
  t_correct<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, u, tr->var[V_DP]);
  KCHECK ();
  return bc_leaves (tr, u);
}

// gfs_poisson_solve, src/poisson.c:1225-1269
int poisson_solve (gfship_tree * tr, gfship_multilevel_params * par, double * lhs, const double * rhs, double dt)
{
  int e;
  const unsigned minlevel = par->minlevel;
  par->depth = tr->H.depth;
  par->niter = 0;
  if ((e = residual_norm (tr, lhs, rhs, tr->var[V_RES], dt, &par->residual))) return e;
  par->residual_before = par->residual;
  double res_max_before = par->residual.infty;
  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    if ((e = poisson_cycle (tr, par, lhs, rhs))) return e;
    if ((e = residual_norm (tr, lhs, rhs, tr->var[V_RES], dt, &par->residual))) return e;
    if (par->residual.infty == res_max_before)
      break;
    if (par->residual.infty > res_max_before/1.1 && par->minlevel < par->depth)
      par->minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }
  par->minlevel = minlevel;
  return 0;
}

P3 p3 (gfship_tree * tr, int first)
{
  P3 a;
  for (int c = 0; c < 3; c++) a.p[c] = tr->var[first + c];
  return a;
}

P6 p6 (gfship_tree * tr, int first)
{
  P6 a;
  for (int d = 0; d < 6; d++) a.p[d] = tr->var[first + d];
  return a;
}

int gather_un (gfship_tree * tr, int kind, int dmask)
{
  FaceSet & F = tr->fs[kind];
  t_gather_un<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, F.faces, F.inc_off, F.inc, F.fval,
      p6 (tr, V_UN), dmask);
  KCHECK ();
  return 0;
}

// mac_projection, src/timestep.c:356-444
int mac_projection (gfship_tree * tr, gfship_multilevel_params * par, double dt, double * p, int gvar)
{
  int e;
  const int dim = tr->H.dim;
  for (int c = 0; c < dim; c++)   /* gfs_reset_gradients on the leaves (the other cells are never read) */
    GFSHIP_HIP (hipMemsetAsync (tr->var[gvar + c], 0, tr->ncell*sizeof (double), tr->stream));
  t_divergence<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, p6 (tr, V_UN), tr->var[V_DIV], dt);
  KCHECK ();
  if ((e = poisson_solve (tr, par, p, tr->var[V_DIV], dt))) return e;
  // gfs_correct_normal_velocities, src/timestep.c:163-179: FTT_XY (the x faces, then the y faces) in
  // 2-D, FTT_XYZ (one traversal) in 3-D
  for (int k = 0; k < (dim == 2 ? 2 : 1); k++) {
    FaceSet & F = tr->fs[dim == 2 ? 1 + k : 0];
    P3 gv = { { nullptr, nullptr, nullptr } };
    if (dim == 2) gv.p[k] = tr->var[gvar + k];
    else gv = p3 (tr, gvar);
    t_face_correct<<<blocks (F.nfaces), 256, 0, tr->stream>>> (tr->D, F.faces, F.nfaces, p, F.fval);
    KCHECK ();
    t_gather_correct<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, F.faces, F.inc_off, F.inc, F.fval,
	p6 (tr, V_UN), gv, dt);
    KCHECK ();
  }
  // gfs_scale_gradients, src/timestep.c:92-107
  t_scale<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, p3 (tr, gvar), 0., 0, p3 (tr, gvar));
  KCHECK ();
  for (int c = 0; c < dim; c++)
    if ((e = bc_leaves (tr, tr->var[gvar + c], c))) return e;
  return 0;
}

int correct_centered (gfship_tree * tr, int gvar, double dt)   /* src/timestep.c:498-530 */
{
  int e;
  t_scale<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, p3 (tr, V_U), dt, 1, p3 (tr, gvar));
  KCHECK ();
  for (int c = 0; c < tr->H.dim; c++)
    if ((e = bc_velocity (tr, c))) return e;
  return 0;
}

int approximate_projection (gfship_tree * tr, gfship_multilevel_params * par, double dt)
{ /* src/timestep.c:560-596 */
  int e;
  FaceSet & F = tr->fs[0];
  t_face_interp<<<blocks (F.nfaces), 256, 0, tr->stream>>> (tr->D, F.faces, F.nfaces, p3 (tr, V_U), F.fval);
  KCHECK ();
  if ((e = gather_un (tr, 0, 63))) return e;
  if ((e = mac_projection (tr, par, dt, tr->var[V_P], V_G))) return e;
  return correct_centered (tr, V_G, dt);
}

UpwindArgs upwind_args (gfship_tree * tr)
{
  UpwindArgs A;
  for (int c = 0; c < 3; c++) A.u[c] = tr->var[V_U + c];
  for (int d = 0; d < 6; d++) { A.un[d] = tr->var[V_UN + d]; A.fv[d] = tr->var[V_FV + d]; }
  return A;
}

int face_values_set (gfship_tree * tr, const double * v, double dt, int use_centered, int comp, int gradient = 0)
{ /* src/timestep.c:644-654 */
  AdvArgs A;
  A.v = v; A.dt = dt; A.use_centered = use_centered; A.gradient = gradient;
  A.visc = A.gsrc = 0.;
  for (int c = 0; c < 3; c++) if (v == tr->var[V_U + c]) { A.visc = tr->visc[c]; A.gsrc = tr->src[c]; }
  for (int c = 0; c < 3; c++) A.u[c] = tr->var[V_U + c];
  for (int d = 0; d < 6; d++) { A.un[d] = tr->var[V_UN + d]; A.fv[d] = tr->var[V_FV + d]; }
  t_face_values<<<blocks (tr->nleaves*tr->H.dim), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, A);
  KCHECK ();
  if (tr->nghost_leaves)
  {
    Sgn6 wall;
    int ucomp = -1;
    for (int c = 0; c < 3; c++) if (v == tr->var[V_U + c]) ucomp = c;
    for (int d = 0; d < 6; d++)
      wall.s[d] = tr->side[d] != GFSHIP_SIDE_PERIODIC ? 1. + (ucomp >= 0 ? tr->bc_u[ucomp][d] : GFSHIP_BC_SYMMETRY) : 0.;
    t_face_bc<<<blocks (tr->nghost_leaves), 256, 0, tr->stream>>> (tr->ghost_leaves, tr->nghost_leaves, p6 (tr, V_FV), wall, comp,
	v, ucomp >= 0 ? tr->var[V_BCU + ucomp] : nullptr);
  }
  KCHECK ();
  return 0;
}

// gfs_diffusion_residual + the norm of gfs_diffusion (src/timestep.c:756-760)
int diffusion_residual_norm (gfship_tree * tr, const double * u, const double * rhs, double w, gfship_norm * out)
{
  GFSHIP_HIP (hipMemsetAsync (tr->d_red, 0, 5*sizeof (double), tr->stream));
  t_diffusion_residual<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, u, rhs,
      tr->var[V_RES], w, tr->d_red);
  KCHECK ();
  GFSHIP_HIP (hipMemcpyAsync (tr->h_red, tr->d_red, 5*sizeof (double), hipMemcpyDeviceToHost, tr->stream));
  GFSHIP_HIP (hipStreamSynchronize (tr->stream));
  const double ws = tr->h_red[4];
  out->w = ws;
  out->infty = ws > 0. ? tr->h_red[0] : 0.;
  out->bias = ws > 0. ? tr->h_red[1]/ws : 0.;
  out->first = ws > 0. ? tr->h_red[2]/ws : 0.;
  out->second = ws > 0. ? sqrt (tr->h_red[3]/ws) : 0.;
  return 0;
}

// gfs_diffusion_cycle, src/poisson.c:1633-1690: the residual restricted by gfs_get_from_below_intensive,
// 10 nrelax sweeps on the first level, the homogeneous conditions of the variable in the loops
int diffusion_cycle (gfship_tree * tr, unsigned levelmin, unsigned depth, unsigned nrelax, int c, double w)
{
  int e;
  const Sgn6 sg = homogeneous_signs_u (tr, c);
  double * u = tr->var[V_U + c];
  if ((e = from_below (tr, tr->var[V_RES], 0))) return e;
  GFSHIP_HIP (hipMemsetAsync (tr->var[V_DP], 0, tr->ncell*sizeof (double), tr->stream));
  if ((e = relax_loop (tr, levelmin, 10*nrelax, 1., 1, w, &sg))) return e;
  for (unsigned m = levelmin + 1; m <= depth; m++) {
    if (tr->nnonleaf[m - 1]) {
      t_from_above<<<blocks (tr->nnonleaf[m - 1]), 256, 0, tr->stream>>> (tr->D, tr->nonleaf[m - 1], tr->nnonleaf[m - 1], tr->var[V_DP]);
      KCHECK ();
    }
    if ((e = relax_loop (tr, m, nrelax, 1., 1, w, &sg))) return e;
  }
  t_correct<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, u, tr->var[V_DP]);
  KCHECK ();
  return bc_velocity (tr, c);
}

// variable_diffusion (src/timestep.c:923-949) + gfs_diffusion (:735-788) of velocity component c; the
// right-hand side is V_DRHS.  Quadtrees: every face carries the weight beta dt D (face_gradient_w)
int variable_diffusion (gfship_tree * tr, int c)
{
  int e;
  gfship_multilevel_params * par = &tr->diffusion_params[c];
  const double w = 1.*(par->beta*tr->dt)*tr->visc[c]*1./1.;      /* diffusion_coef, src/poisson.c:1280-1285 */
  double * v = tr->var[V_U + c], * rhs = tr->var[V_DRHS];
  t_diffusion_rhs<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, v, rhs, w,
      (1. - par->beta)/par->beta);
  KCHECK ();
  unsigned minlevel = par->minlevel;
  const unsigned maxlevel = tr->H.depth;
  if ((e = diffusion_residual_norm (tr, v, rhs, w, &par->residual))) return e;
  par->residual_before = par->residual;
  double res_max_before = par->residual.infty;
  par->niter = 0;
  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    if ((e = diffusion_cycle (tr, minlevel, maxlevel, par->nrelax, c, w))) return e;
    if ((e = diffusion_residual_norm (tr, v, rhs, w, &par->residual))) return e;
    if (par->residual.infty == res_max_before)
      break;
    if (par->residual.infty > res_max_before/1.1 && minlevel < maxlevel)
      minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }
  return 0;
}

int predicted_face_velocities (gfship_tree * tr)   /* src/timestep.c:681-717 */
{
  int e;
  for (int c = 0; c < tr->H.dim; c++) {
    if ((e = face_values_set (tr, tr->var[V_U + c], tr->dt, 1, c))) return e;
    FaceSet & F = tr->fs[1 + c];
    t_face_advected_un<<<blocks (F.nfaces), 256, 0, tr->stream>>> (tr->D, F.faces, F.nfaces, upwind_args (tr), F.fval);
    KCHECK ();
    if ((e = gather_un (tr, 1 + c, 3 << (2*c)))) return e;
  }
  return 0;
}

// gfs_centered_velocity_advection_diffusion, src/timestep.c:976-1016, with variable_sources :872-921
int centered_velocity_advection (gfship_tree * tr, int gmac, int g)
{
  int e;
  FaceSet & F = tr->fs[0];
  for (int c = 0; c < tr->H.dim; c++) {
    if ((e = face_values_set (tr, tr->var[V_U + c], tr->dt, 0, c))) return e;
    t_face_flux<<<blocks (F.nfaces), 256, 0, tr->stream>>> (tr->D, F.faces, F.nfaces, upwind_args (tr), tr->var[gmac + c], tr->dt, F.fval);
    KCHECK ();
    double * sv = tr->var[V_U + c];
    if (tr->visc[c] != 0.) {
      /* source_diffusion (v[c]): rhs = copy of v, the sources into rhs, then the implicit solve
	 (src/timestep.c:996-1007) */
      sv = tr->var[V_DRHS];
      t_copy_leaves<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, tr->var[V_U + c], sv);
      KCHECK ();
    }
    t_gather_flux<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, F.faces, F.inc_off, F.inc, F.fval,
	sv, tr->var[g + c], tr->dt, tr->src[c]);
    KCHECK ();
    if (tr->visc[c] != 0. && (e = variable_diffusion (tr, c))) return e;
  }
  for (int c = 0; c < tr->H.dim; c++)
    if ((e = bc_velocity (tr, c))) return e;
  return 0;
}

// a scalar with the default GfsBc on the GfsBoundary sides (symmetry, src/boundary.c:45-62: the ghost
// takes the value of the cell it touches), the periodic image elsewhere
int bc_scalar (gfship_tree * tr, double * v)
{
  if (!tr->nghost_leaves) return 0;
  Sgn6 sg;
  for (int d = 0; d < 6; d++) sg.s[d] = 1.;
  t_copy_ghosts_signed<<<blocks (tr->nghost_leaves), 256, 0, tr->stream>>> (tr->ghost_leaves, tr->nghost_leaves, v, sg);
  KCHECK ();
  return 0;
}

// gfs_tracer_advection_diffusion (src/timestep.c:1028-1055, no diffusion) with variable_sources
// :872-921: the face values with the gradient of the GfsVariableTracer, gfs_face_advection_flux, the
// update, gfs_domain_bc
int tracer_advection (gfship_tree * tr, int k, double dt)
{
  int e;
  FaceSet & F = tr->fs[0];
  double * v = tr->var[V_T + k];
  if ((e = face_values_set (tr, v, dt, 0, -1, tr->tracer_gradient[k]))) return e;
  t_face_flux<<<blocks (F.nfaces), 256, 0, tr->stream>>> (tr->D, F.faces, F.nfaces, upwind_args (tr), nullptr, dt, F.fval);
  KCHECK ();
  t_gather_flux<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, F.faces, F.inc_off, F.inc, F.fval,
      v, nullptr, dt);
  KCHECK ();
  return bc_scalar (tr, v);
}

int advance_tracers (gfship_tree * tr, double dt)   /* src/simulation.c:405-430 */
{
  int e;
  for (int k = 0; k < tr->ntracers; k++)
    if ((e = tracer_advection (tr, k, dt))) return e;
  return 0;
}

int domain_cfl (gfship_tree * tr, double * cfl)   /* src/domain.c:2899-2923 */
{
  const double big = DBL_MAX;
  GFSHIP_HIP (hipMemcpyAsync (tr->d_red, &big, sizeof (double), hipMemcpyHostToDevice, tr->stream));
  FaceSet & F = tr->fs[0];
  t_cfl_faces<<<blocks (F.nfaces), 256, 0, tr->stream>>> (tr->D, F.faces, F.nfaces, p6 (tr, V_UN), tr->d_red);
  KCHECK ();
  D3 visc = { { tr->visc[0], tr->visc[1], tr->visc[2] } }, srcs = { { tr->src[0], tr->src[1], tr->src[2] } };
  t_cfl_cells<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, p3 (tr, V_U), tr->d_red, visc, srcs);
  KCHECK ();
  GFSHIP_HIP (hipMemcpyAsync (tr->h_red, tr->d_red, sizeof (double), hipMemcpyDeviceToHost, tr->stream));
  GFSHIP_HIP (hipStreamSynchronize (tr->stream));
  *cfl = sqrt (tr->h_red[0]);
  return 0;
}

int set_timestep (gfship_tree * tr)   /* src/simulation.c:1569-1633; the only event time is `end' */
{
  int e;
  double c;
  if ((e = domain_cfl (tr, &c))) return e;
  const double t = tr->t;
  tr->dt = tr->cfl*c;
  double tnext = 2147483647;
  if (tr->next_event)          /* the gfs_event_next loop, src/simulation.c:1603-1610 */
    tnext = tr->next_event (tr->next_event_ctx, t, tr->iter);
  if (tr->end < tnext)
    tnext = tr->end;
  const double n = ceil ((tnext - t)/tr->dt);
  if (n > 0. && n < 2147483647) {
    tr->dt = (tnext - t)/n;
    if (n == 1.)
      tr->tnext = tnext;
    else
      tr->tnext = t + tr->dt;
  }
  else
    tr->tnext = t + tr->dt;
  if (tr->dt < 1e-9)
    tr->dt = 1e-9;
  return 0;
}

int coarse_init (gfship_tree * tr)   /* src/adaptive.c:43-58 */
{
  const int vars[] = { V_P, V_PMAC, V_U, V_U + 1, V_U + 2 };
  PN V;
  int nv = 0;
  for (int k = 0; k < 2 + tr->H.dim; k++) V.p[nv++] = tr->var[vars[k]];
  for (int k = 0; k < tr->ntracers && nv < 8; k++) V.p[nv++] = tr->var[V_T + k];
  return from_below_n (tr, V, nv);
}

void tree_free (gfship_tree * tr)
{
  if (!tr) return;
  (void) hipFree (tr->dflag);
  (void) hipFree (tr->d_nbtab); (void) hipFree (tr->d_child0); (void) hipFree (tr->d_cmask); (void) hipFree (tr->d_idtab);
  for (double * p : tr->var) (void) hipFree (p);
  (void) hipFree (tr->leaves);
  (void) hipFree (tr->ghost_leaves);
  for (int l = 0; l <= GFSHIP_MAXLEVEL; l++) {
    (void) hipFree (tr->nonleaf[l]);
    (void) hipFree (tr->sweep[l].cells); (void) hipFree (tr->sweep[l].lev_off); (void) hipFree (tr->sweep[l].ghosts);
    (void) hipFree (tr->sweep[l].ti); (void) hipFree (tr->sweep[l].td); (void) hipFree (tr->sweep[l].tv);
    (void) hipFree (tr->sweep[l].cell_off); (void) hipFree (tr->sweep[l].chunk);
    loop_free (tr->sweep[l].loop);
    for (Sweep::Loop & P : tr->sweep[l].dloop) loop_free (P);
  }
  for (FaceSet & F : tr->fs) { (void) hipFree (F.faces); (void) hipFree (F.inc_off); (void) hipFree (F.inc); (void) hipFree (F.fval); }
  (void) hipFree (tr->d_red);
  if (tr->h_red) (void) hipHostFree (tr->h_red);
  if (tr->stream) (void) hipStreamDestroy (tr->stream);
  delete tr;
}

} // namespace

// ---- C ABI -------------------------------------------------------------------------------------

extern "C" {

int gfship_tree_create (gfship_tree ** out, int dim, gfship_refine_fn refine, void * ctx, int device)
{
  return gfship_tree_create_sides (out, dim, refine, ctx, nullptr, device);
}

static int tree_create_sides (gfship_tree ** out, int dim, gfship_refine_fn refine, void * ctx,
			      const int * side, int device);

// no exception crosses the C ABI: a tree too large for the host comes back as an error
int gfship_tree_create_sides (gfship_tree ** out, int dim, gfship_refine_fn refine, void * ctx,
			      const int * side, int device)
{
  try {
    return tree_create_sides (out, dim, refine, ctx, side, device);
  }
  catch (const std::bad_alloc &) {
    set_error ("gfship_tree_create: out of host memory while the tree was built");
    return GFSHIP_EUNSUPPORTED;
  }
}

static int tree_create_sides (gfship_tree ** out, int dim, gfship_refine_fn refine, void * ctx,
			      const int * side, int device)
{
  GFSHIP_CHECK (out && refine, GFSHIP_EINVAL, "gfship_tree_create: null argument");
  GFSHIP_CHECK (dim == 2 || dim == 3, GFSHIP_EINVAL, "gfship_tree_create: dim = %d", dim);
  if (!g_host_only) {
    int ndev = 0;
    if (hipGetDeviceCount (&ndev) != hipSuccess || ndev == 0) {
      set_error ("gfship_tree_create: no HIP device");
      return GFSHIP_ENODEVICE;
    }
    GFSHIP_CHECK (device >= 0 && device < ndev, GFSHIP_EINVAL, "gfship_tree_create: device %d of %d", device, ndev);
    GFSHIP_HIP (hipSetDevice (device));
  }

  // gfs_refine_refine + gfs_simulation_refine, src/refine.c:45-60, src/simulation.c:1203-1233
  Builder B;
  B.dim = dim;
  B.ensure (0);
  B.flag[0][B.idx (0, 1, 1, 1)] = LEAF;
  int e = B.refine_rec (0, 1, 1, 1, refine, ctx);
  GFSHIP_CHECK (!B.too_large, GFSHIP_EUNSUPPORTED,
		"gfship_tree_create: more than 2^27 cells in the dense levels of the tree");
  GFSHIP_CHECK (e == 0, e, "gfship_tree_create: more than %d levels", GFSHIP_MAXLEVEL);
  gfship_tree * tr = new gfship_tree;
  tr->device = device;
  for (int d = 0; d < 2*dim; d++) {
    tr->side[d] = side ? side[d] : GFSHIP_SIDE_PERIODIC;
    if (tr->side[d] != GFSHIP_SIDE_PERIODIC && tr->side[d] != GFSHIP_SIDE_BOUNDARY) {
      delete tr;
      set_error ("gfship_tree_create: side %d: periodic or boundary", d);
      return GFSHIP_EINVAL;
    }
    if (tr->side[d] == GFSHIP_SIDE_BOUNDARY) tr->has_boundary = true;
    tr->bc_p[d] = GFSHIP_BC_SYMMETRY;
  }
  {
    long long total = 0;      /* 32-bit cell indices, times the number of directions in the tables */
    for (size_t l = 0; l < B.flag.size (); l++)
      total += (long long) B.flag[l].size ();
    if (total > (1ll << 27)) {
      delete tr;
      set_error ("gfship_tree_create: more than 2^27 cells in the dense levels of the tree");
      return GFSHIP_EUNSUPPORTED;
    }
  }
  auto flatten = [&] () {
    Topo & H = tr->H;
    H.dim = dim;
    H.depth = (int) B.flag.size () - 1;
    int off = 0;
    for (int l = 0; l <= H.depth; l++) {
      H.off[l] = off;
      off += H.lsize (l);
    }
    H.off[H.depth + 1] = off;
    tr->ncell = off;
    tr->hflag.assign (off, NONE);
    for (int l = 0; l <= H.depth; l++)
      std::copy (B.flag[l].begin (), B.flag[l].end (), tr->hflag.begin () + H.off[l]);
    H.flag = tr->hflag.data ();
  };
  flatten ();
  for (int l = tr->H.depth - 2; l >= 0; l--) {
    // the refinements of a level are applied while the level is traversed (simulation.c:1105-1109)
    traverse (tr->H, root_cell (tr->H), T_LEVEL, l, [&] (Cell c) {
	if (tr->H.leaf (c) && refine_corner (tr->H, c)) {
	  (void) B.refine_single (c.l, tr->H.ci (c), tr->H.cj (c), tr->H.ck (c));
	  for (int ll = 0; ll <= tr->H.depth; ll++)
	    std::copy (B.flag[ll].begin (), B.flag[ll].end (), tr->hflag.begin () + tr->H.off[ll]);
	}
      });
  }
  // gfs_domain_match: the ghost trees of the periodic sides mirror the cells they face; both cells
  // of a periodic pair must be at the same refinement
  {
    std::vector<Ghost> all;
    Topo every = tr->H;
    for (int l = 0; l <= tr->H.depth; l++) {
      // every ghost position of the level, whatever its flag: mark them present for the listing
      const int n = 1 << l, r = n + 2;
      unsigned char * f = tr->hflag.data () + tr->H.off[l];
      for (int side = 0; side < 2*dim; side++)
	for (int tb = 1; tb <= (dim == 3 ? n : 1); tb++)
	  for (int ta = 1; ta <= n; ta++) {
	    int g[3], im[3], own[3];
	    const int a = side/2, o1 = a == 0 ? 1 : 0, o2 = a == 2 ? 1 : 2;
	    g[a] = (side & 1) ? 0 : n + 1;
	    im[a] = (side & 1) ? n : 1;
	    own[a] = (side & 1) ? 1 : n;
	    g[o1] = im[o1] = own[o1] = ta;
	    g[o2] = im[o2] = own[o2] = dim == 3 ? tb : 0;
	    const int G = g[0] + r*(g[1] + r*g[2]), I = im[0] + r*(im[1] + r*im[2]),
	      O = own[0] + r*(own[1] + r*own[2]);
	    if (tr->side[side] != GFSHIP_SIDE_PERIODIC) {    /* the ghost tree of a GfsBoundary matches its side */
	      f[G] = f[O];
	      continue;
	    }
	    if (f[O] != f[I]) {
	      delete tr;
	      set_error ("gfship_tree_create: the refinement differs across a periodic side (level %d)", l);
	      return GFSHIP_EUNSUPPORTED;
	    }
	    f[G] = f[I];
	  }
    }
    (void) every; (void) all;
  }

  const Topo & T = tr->H;
  if (!g_host_only) {
    hipError_t he = hipStreamCreate (&tr->stream);
    if (he != hipSuccess) { tree_free (tr); return hip_fail (he, "hipStreamCreate", __FILE__, __LINE__); }
  }
#define TRY(call) do { int e_ = (call); if (e_) { tree_free (tr); return e_; } } while (0)
#define TRYHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { tree_free (tr); \
      return hip_fail (e_, #call, __FILE__, __LINE__); } } while (0)
  if (!g_host_only) {
    TRYHIP (hipMalloc ((void **) &tr->dflag, tr->ncell));
    TRYHIP (hipMemcpy (tr->dflag, tr->hflag.data (), tr->ncell, hipMemcpyHostToDevice));
  }
  {
    // the tables of Topo (tree.hpp), from the computed answers
    Topo & H = tr->H;
    const int nd = H.nd ();
    tr->h_nbtab.assign ((size_t) tr->ncell*nd, -1);
    tr->h_child0.assign (tr->ncell, 0);
    tr->h_cmask.assign (tr->ncell, 0);
    tr->h_idtab.assign (tr->ncell, 0);
    for (int l = 0; l <= H.depth; l++)
      for (int q = 0; q < H.lsize (l); q++) {
	const int g = H.off[l] + q;
	if (tr->hflag[g] == NONE)
	  continue;
	const Cell c = { l, q };
	for (int d = 0; d < nd; d++) {
	  const Cell nb = H.neighbor (c, d);
	  tr->h_nbtab[(size_t) g*nd + d] = !exists (nb) ? -1 : nb.l == l ? nb.q : - nb.q - 2;
	}
	tr->h_idtab[g] = (unsigned char) (H.id (c) | (H.interior (c) ? 8 : 0));
	if (l < H.depth) {
	  const int rr = H.r (l + 1);
	  tr->h_child0[g] = 2*H.ci (c) - 1 + rr*(2*H.cj (c) + (dim == 3 ? rr*2*H.ck (c) : 0));
	  for (int k = 0; k < H.nc (); k++)
	    if (exists (H.child (c, k)))
	      tr->h_cmask[g] |= (unsigned char) (1 << k);
	}
      }
    H.nbtab = tr->h_nbtab.data ();
    H.child0 = tr->h_child0.data ();
    H.cmask = tr->h_cmask.data ();
    H.idtab = tr->h_idtab.data ();
  }
  TRY (to_device (tr->h_nbtab, &tr->d_nbtab));
  TRY (to_device (tr->h_child0, &tr->d_child0));
  TRY (to_device (tr->h_cmask, &tr->d_cmask));
  TRY (to_device (tr->h_idtab, &tr->d_idtab));
  tr->D = tr->H;
  tr->D.flag = tr->dflag;
  tr->D.nbtab = tr->d_nbtab;
  tr->D.child0 = tr->d_child0;
  tr->D.cmask = tr->d_cmask;
  tr->D.idtab = tr->d_idtab;
  for (int v = 0; v < V_NVAR && !g_host_only; v++) {
    TRYHIP (hipMalloc ((void **) &tr->var[v], tr->ncell*sizeof (double)));
    TRYHIP (hipMemset (tr->var[v], 0, tr->ncell*sizeof (double)));
  }
  if (!g_host_only) {
    TRYHIP (hipMalloc ((void **) &tr->d_red, 8*sizeof (double)));
    TRYHIP (hipHostMalloc ((void **) &tr->h_red, 8*sizeof (double), 0));
  }
  traverse (T, root_cell (T), T_LEAFS, -1, [&] (Cell c) { tr->hleaves.push_back (c); });
  tr->nleaves = (int) tr->hleaves.size ();
  TRY (to_device (tr->hleaves, &tr->leaves));
  for (int l = 0; l < T.depth; l++) {
    std::vector<Cell> nl;
    traverse (T, root_cell (T), T_LEVEL_NON_LEAFS, l, [&] (Cell c) { nl.push_back (c); });
    tr->nnonleaf[l] = (int) nl.size ();
    TRY (to_device (nl, &tr->nonleaf[l]));
  }
  std::vector<Ghost> gh;
  ghost_list (T, tr->side, T_LEAFS, -1, gh);
  tr->nghost_leaves = (int) gh.size ();
  TRY (to_device (gh, &tr->ghost_leaves));
  for (int m = 0; m <= T.depth; m++)
    TRY (sweep_plan (tr, m, &tr->sweep[m]));
  for (int k = 0; k < 1 + dim; k++)
    TRY (face_set (tr, k, &tr->fs[k]));
#undef TRY
#undef TRYHIP
  gfship_multilevel_params_init (&tr->projection_params, dim);
  gfship_multilevel_params_init (&tr->approx_projection_params, dim);
  for (int c = 0; c < 3; c++) {     /* diffusion_init, src/source.c:966-974 */
    gfship_multilevel_params_init (&tr->diffusion_params[c], dim);
    tr->diffusion_params[c].tolerance = 1e-6;
  }
  *out = tr;
  return GFSHIP_OK;
}

void gfship_tree_destroy (gfship_tree * tr) { tree_free (tr); }

int gfship_tree_depth (const gfship_tree * tr) { return tr ? tr->H.depth : -1; }
int gfship_tree_dim (const gfship_tree * tr) { return tr ? tr->H.dim : -1; }

int gfship_tree_flags (const gfship_tree * tr, int level, unsigned char * out)
{
  GFSHIP_CHECK (tr && out && level >= 0 && level <= tr->H.depth, GFSHIP_EINVAL, "gfship_tree_flags: bad argument");
  memcpy (out, tr->hflag.data () + tr->H.off[level], (size_t) tr->H.lsize (level));
  return GFSHIP_OK;
}

int gfship_tree_upload (gfship_tree * tr, int var, int level, const double * in)
{
  GFSHIP_CHECK (tr && in && var >= 0 && var < abi_nvar && level >= 0 && level <= tr->H.depth, GFSHIP_EINVAL,
		"gfship_tree_upload: bad argument");
  GFSHIP_HIP (hipMemcpyAsync (tr->var[abi_var[var]] + tr->H.off[level], in, (size_t) tr->H.lsize (level)*sizeof (double),
			      hipMemcpyHostToDevice, tr->stream));
  GFSHIP_HIP (hipStreamSynchronize (tr->stream));
  return GFSHIP_OK;
}

int gfship_tree_download (gfship_tree * tr, int var, int level, double * out)
{
  GFSHIP_CHECK (tr && out && var >= 0 && var < abi_nvar && level >= 0 && level <= tr->H.depth, GFSHIP_EINVAL,
		"gfship_tree_download: bad argument");
  GFSHIP_HIP (hipMemcpyAsync (out, tr->var[abi_var[var]] + tr->H.off[level], (size_t) tr->H.lsize (level)*sizeof (double),
			      hipMemcpyDeviceToHost, tr->stream));
  GFSHIP_HIP (hipStreamSynchronize (tr->stream));
  return GFSHIP_OK;
}

gfship_multilevel_params * gfship_tree_projection_params (gfship_tree * tr, int approx)
{
  return approx ? &tr->approx_projection_params : &tr->projection_params;
}

int gfship_tree_set_bc (gfship_tree * tr, int d, int kind)
{
  GFSHIP_CHECK (tr && d >= 0 && d < 2*tr->H.dim && tr->side[d] == GFSHIP_SIDE_BOUNDARY &&
		(kind == GFSHIP_BC_SYMMETRY || kind == GFSHIP_BC_DIRICHLET || kind == GFSHIP_BC_NEUMANN),
		GFSHIP_EINVAL, "gfship_tree_set_bc: bad argument");
  tr->bc_p[d] = kind;
  for (int l = 0; l <= GFSHIP_MAXLEVEL; l++)     /* the signs of the homogeneous conditions are in the plans */
    loop_free (tr->sweep[l].loop);
  return GFSHIP_OK;
}

/* gfs_poisson_solve (src/poisson.c:1225-1269) on the tree: P (GFSHIP_TREE_P) holds the guess, the
   right-hand side is in GFSHIP_TREE_DIV, dia = 0, alpha = NULL; the residual is left in GFSHIP_TREE_RES */
int gfship_tree_poisson_solve (gfship_tree * tr, gfship_multilevel_params * par, double dt)
{
  GFSHIP_CHECK (tr && par, GFSHIP_EINVAL, "gfship_tree_poisson_solve: null argument");
  GFSHIP_HIP (hipSetDevice (tr->device));
  int e;
  if ((e = bc_leaves (tr, tr->var[V_P]))) return e;
  return poisson_solve (tr, par, tr->var[V_P], tr->var[V_DIV], dt);
}

int gfship_tree_set_time (gfship_tree * tr, double end, double cfl)
{
  GFSHIP_CHECK (tr && cfl > 0., GFSHIP_EINVAL, "gfship_tree_set_time: bad argument");
  tr->end = end;
  tr->cfl = cfl;
  return GFSHIP_OK;
}

int gfship_tree_set_next_event (gfship_tree * tr, gfship_next_event_fn fn, void * ctx)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_set_next_event: null tree");
  tr->next_event = fn;
  tr->next_event_ctx = ctx;
  return GFSHIP_OK;
}

double gfship_tree_time (const gfship_tree * tr) { return tr->t; }
double gfship_tree_dt (const gfship_tree * tr) { return tr->dt; }
unsigned gfship_tree_iter (const gfship_tree * tr) { return tr->iter; }

/* gfs_divergence of (U, V, W) on the leaves into the variable GFSHIP_TREE_DIV (the scratch `div' of
   the projections: valid until the next step) */
int gfship_tree_divergence (gfship_tree * tr)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_divergence: null tree");
  GFSHIP_HIP (hipSetDevice (tr->device));
  GFSHIP_HIP (hipMemsetAsync (tr->var[V_DIV], 0, tr->ncell*sizeof (double), tr->stream));
  t_divergence_centered<<<blocks (tr->nleaves), 256, 0, tr->stream>>> (tr->D, tr->leaves, tr->nleaves, p3 (tr, V_U), tr->var[V_DIV]);
  KCHECK ();
  return GFSHIP_OK;
}

/* Host-side self-check of the plans of a tree, without a device (for the CPU tests): builds the tree
   and, for every level m, runs the relax loop of nrelax sweeps on pseudo-random values three ways on
   the host -- (1) the reference's program: copies of the ghosts, the cells in tree order through the
   stencil code that walks the tree (tree.hpp), sweep after sweep; (2) sweep after sweep through the
   compiled stencils, the cells of a dependency level in REVERSE order; (3) the plan of the whole loop,
   the nodes of a level in reverse order -- and compares the three results bit for bit.
   stats[0] = cells of all sweeps, [1] = dependency levels sweep after sweep, [2] = levels of the loop
   plans, [3] = cells whose results differ (0 when the plans are right). */
int gfship_tree_host_check (int dim, gfship_refine_fn refine, void * ctx, const int * side,
			    unsigned nrelax, long long stats[4])
{
  GFSHIP_CHECK (refine && stats && nrelax >= 1, GFSHIP_EINVAL, "gfship_tree_host_check: bad argument");
  gfship_tree * tr = nullptr;
  g_host_only = true;
  int e = gfship_tree_create_sides (&tr, dim, refine, ctx, side, 0);
  if (e) { g_host_only = false; return e; }
  const Topo & T = tr->H;
  stats[0] = stats[1] = stats[2] = stats[3] = 0;
  const Sgn6 sg = homogeneous_signs (tr);
  struct HostReader {
    const double * p;
    double operator() (const Topo & T, Cell c) const { return p[T.gi (c)]; }
  };
  for (int m = 0; m <= T.depth && !e; m++) {
    Sweep & S = tr->sweep[m];
    e = loop_plan (tr, m, nrelax, &S);
    if (e) break;
    std::vector<double> u0 (tr->ncell), rhs (tr->ncell);
    unsigned long long seed = 88172645463325252ull + m;
    for (int g = 0; g < tr->ncell; g++) {
      seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
      u0[g] = (double) (seed % 2000001)/1e6 - 1.;
      seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
      rhs[g] = (double) (seed % 2000001)/1e6 - 1.;
    }
    const double omega = 1.;
    // (1) the sequential program through the tree-walking stencil code
    std::vector<double> a = u0;
    std::vector<Cell> order;
    traverse (T, root_cell (T), T_LEVEL_LEAFS, m, [&] (Cell c) { order.push_back (c); });
    for (unsigned sw = 0; sw < nrelax; sw++) {
      for (const Ghost & G : S.h_ghosts) a[G.g] = sg.s[G.side]*a[G.img];
      HostReader R = { a.data () };
      for (Cell c : order) {
	const int g = T.gi (c);
	a[g] = relax_cell (T, c, R, rhs[g], omega, m);
      }
    }
    // (2) compiled stencils, sweep after sweep, each dependency level backwards
    std::vector<double> b = u0;
    {
      std::vector<int> lev_of (S.h_g.size (), 0);   /* recompute the levels from the read sets */
      std::vector<int> wl (tr->ncell, 0), rl (tr->ncell, 0);
      int nl = 0;
      for (size_t c = 0; c < S.h_g.size (); c++) {
	const int w = S.h_g[c];
	int L = std::max (rl[w], wl[w]);
	for (int k = S.h_cell_off[3*c + 2]; k < S.h_cell_off[3*(c + 1) + 2]; k++) L = std::max (L, wl[S.h_tv[k]]);
	L++;
	for (int k = S.h_cell_off[3*c + 2]; k < S.h_cell_off[3*(c + 1) + 2]; k++) rl[S.h_tv[k]] = std::max (rl[S.h_tv[k]], L);
	wl[w] = L; rl[w] = 0;
	lev_of[c] = L; nl = std::max (nl, L);
      }
      stats[1] += (long long) nl*nrelax;
      for (unsigned sw = 0; sw < nrelax; sw++) {
	for (const Ghost & G : S.h_ghosts) b[G.g] = sg.s[G.side]*b[G.img];
	for (int L = 1; L <= nl; L++) {
	  std::vector<std::pair<int, double>> out;
	  for (size_t c = S.h_g.size (); c-- > 0; )
	    if (lev_of[c] == L) {
	      std::vector<double> vals;
	      for (int k = S.h_cell_off[3*c + 2]; k < S.h_cell_off[3*(c + 1) + 2]; k++) vals.push_back (b[S.h_tv[k]]);
	      TapeCursor cur = { S.h_ti.data () + S.h_cell_off[3*c], S.h_td.data () + S.h_cell_off[3*c + 1], vals.data () };
	      const double self = *cur.tv++;
	      double ga, gb;
	      tape_cell (cur, T.nd (), T.dim, T.ncd (), ga, gb);
	      const int g = S.h_g[c];
	      double x = 0.;
	      if (ga != 0.)
		x = T.dim == 2 ? (1. - omega)*self + omega*(gb - rhs[g])/ga : (gb - rhs[g])/ga;
	      out.push_back ({ g, x });
	    }
	  for (auto & kv : out) b[kv.first] = kv.second;      /* the level's stores after its loads */
	}
      }
    }
    // (3) the plan of the whole loop, the nodes of a level backwards
    std::vector<double> c3 = u0;
    {
      const Sweep::Loop & P = S.loop;
      stats[2] += P.nlev;
      size_t k0 = 0;
      const size_t nn = P.h_node_g.size ();
      while (k0 < nn) {
	size_t k1 = k0;
	while (k1 < nn && P.h_node_level[k1] == P.h_node_level[k0]) k1++;
	std::vector<std::pair<int, double>> out;
	for (size_t k = k1; k-- > k0; ) {
	  std::vector<double> vals;
	  for (int q = P.h_node_off[3*k + 2]; q < P.h_node_off[3*(k + 1) + 2]; q++) vals.push_back (c3[P.h_tv[q]]);
	  TapeCursor cur = { P.h_ti.data () + P.h_node_off[3*k], P.h_td.data () + P.h_node_off[3*k + 1], vals.data () };
	  const int g = P.h_node_g[k];
	  double x;
	  if (*cur.ti == K_GHOST)
	    x = (*cur.td)*(*cur.tv);
	  else {
	    const double self = *cur.tv++;
	    double ga, gb;
	    tape_cell (cur, T.nd (), T.dim, T.ncd (), ga, gb);
	    x = 0.;
	    if (ga != 0.)
	      x = T.dim == 2 ? (1. - omega)*self + omega*(gb - rhs[g])/ga : (gb - rhs[g])/ga;
	  }
	  out.push_back ({ g, x });
	}
	for (auto & kv : out) c3[kv.first] = kv.second;
	k0 = k1;
      }
    }
    // (4) the flow plan (tree_flow.hpp) with the kernel's timing of loads and stores
    std::vector<double> c4 = u0;
    if (S.loop.flow) {
      const long long hz = flow_emulate (*S.loop.flow, T.dim, c4, rhs, omega, 0, 1.);
      stats[3] += hz;
      if (getenv ("GFSHIP_TREE_DEBUG")) {
	long long nd4 = 0;
	for (int g = 0; g < tr->ncell; g++) nd4 += memcmp (&a[g], &c4[g], sizeof (double)) != 0;
	fprintf (stderr, "gfship_tree: flow plan of level %d: %lld hazards, %lld values differ\n", m, hz, nd4);
	int widest = 0;
	for (int L = 0; L < S.loop.flow->nlev; L++)
	  widest = std::max (widest, S.loop.flow->h_lev_off[L + 1] - S.loop.flow->h_lev_off[L]);
	fprintf (stderr, "gfship_tree: flow plan of level %d: %d levels, widest %d\n", m, S.loop.flow->nlev, widest);
      }
    }
    else
      c4 = a;
    stats[0] += (long long) order.size ()*nrelax;
    for (int g = 0; g < tr->ncell; g++)
      if (memcmp (&a[g], &b[g], sizeof (double)) || memcmp (&a[g], &c3[g], sizeof (double)) ||
	  memcmp (&a[g], &c4[g], sizeof (double)))
	stats[3]++;
  }
  tree_free (tr);
  g_host_only = false;
  return e;
}

/* the cells of the sweep of level `level' and the number of dependency levels they form */
int gfship_tree_sweep_levels (const gfship_tree * tr, int level, int * ncells, int * nlevels)
{
  GFSHIP_CHECK (tr && level >= 0 && level <= tr->H.depth, GFSHIP_EINVAL, "gfship_tree_sweep_levels: bad argument");
  if (ncells) *ncells = tr->sweep[level].ncells;
  if (nlevels) *nlevels = tr->sweep[level].nlev;
  return GFSHIP_OK;
}

int gfship_tree_set_bc_u (gfship_tree * tr, int c, int d, int kind)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_set_bc_u: null tree");
  GFSHIP_CHECK (c >= 0 && c < tr->H.dim && d >= 0 && d < 2*tr->H.dim, GFSHIP_EINVAL, "component %d / side %d out of range", c, d);
  GFSHIP_CHECK (kind == GFSHIP_BC_SYMMETRY || kind == GFSHIP_BC_DIRICHLET || kind == GFSHIP_BC_NEUMANN, GFSHIP_EINVAL,
		"kind of condition %d", kind);
  GFSHIP_CHECK (tr->side[d] != GFSHIP_SIDE_PERIODIC || kind == GFSHIP_BC_SYMMETRY, GFSHIP_EINVAL, "side %d is periodic", d);
  tr->bc_u[c][d] = kind;
  return GFSHIP_OK;
}

int gfship_tree_set_viscosity (gfship_tree * tr, int c, double nu)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_set_viscosity: null tree");
  GFSHIP_CHECK (c >= 0 && c < tr->H.dim, GFSHIP_EINVAL, "component %d out of range", c);
  GFSHIP_CHECK (nu >= 0., GFSHIP_EINVAL, "the diffusion coefficient must be positive");
  /* octrees: the face coefficients of the coarse side of a fine-coarse face are sums of four quarters that
     may round (src/poisson.c:1280-1303): they would have to be kept per face */
  GFSHIP_CHECK (nu == 0. || tr->H.dim == 2, GFSHIP_EUNSUPPORTED,
		"GfsSourceDiffusion on a refined tree is supported on quadtrees (2-D)");
  tr->visc[c] = nu;
  return GFSHIP_OK;
}

int gfship_tree_set_source (gfship_tree * tr, int c, double g)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_set_source: null tree");
  GFSHIP_CHECK (c >= 0 && c < tr->H.dim, GFSHIP_EINVAL, "component %d out of range", c);
  tr->src[c] = g;
  return GFSHIP_OK;
}

gfship_multilevel_params * gfship_tree_diffusion_params (gfship_tree * tr, int c)
{ return (tr && c >= 0 && c < 3) ? &tr->diffusion_params[c] : nullptr; }

int gfship_tree_add_tracer (gfship_tree * tr, int gradient)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_add_tracer: null tree");
  GFSHIP_CHECK (gradient == 0 || gradient == 1, GFSHIP_EINVAL, "gradient: 0 centred, 1 van Leer");
  GFSHIP_CHECK (tr->ntracers < TREE_MAXTRACERS, GFSHIP_EUNSUPPORTED, "a tree carries %d tracers at most", TREE_MAXTRACERS);
  tr->tracer_gradient[tr->ntracers] = gradient;
  return GFSHIP_TREE_T0 + tr->ntracers++;
}

/* simulation_run up to the loop, src/simulation.c:458-476 */
int gfship_tree_start (gfship_tree * tr)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_start: null tree");
  GFSHIP_HIP (hipSetDevice (tr->device));
  int e;
  const int vars[] = { V_P, V_PMAC, V_U, V_U + 1, V_U + 2 };
  for (int k = 0; k < 2 + tr->H.dim; k++)
    if ((e = k >= 2 ? bc_velocity (tr, k - 2) : bc_leaves (tr, tr->var[vars[k]], -1))) return e;
  for (int k = 0; k < tr->ntracers; k++)
    if ((e = bc_scalar (tr, tr->var[V_T + k]))) return e;
  if ((e = coarse_init (tr))) return e;
  if ((e = set_timestep (tr))) return e;
  if ((e = approximate_projection (tr, &tr->approx_projection_params, tr->dt))) return e;
  if ((e = set_timestep (tr))) return e;
  return advance_tracers (tr, tr->dt/2.);
}

/* one iteration of the loop, src/simulation.c:479-548 */
int gfship_tree_step (gfship_tree * tr)
{
  GFSHIP_CHECK (tr, GFSHIP_EINVAL, "gfship_tree_step: null tree");

  GFSHIP_HIP (hipSetDevice (tr->device));
  int e;
  if ((e = predicted_face_velocities (tr))) return e;
  /* gfs_variables_swap (p, pmac) around the MAC projection */
  if ((e = mac_projection (tr, &tr->projection_params, tr->dt/2., tr->var[V_PMAC], V_GM))) return e;
  const int g = tr->iter > 0 ? V_G : V_GM;
  if ((e = centered_velocity_advection (tr, V_GM, g))) return e;
  if ((e = correct_centered (tr, g, - tr->dt))) return e;
  if ((e = coarse_init (tr))) return e;
  if ((e = approximate_projection (tr, &tr->approx_projection_params, tr->dt))) return e;
  tr->t = tr->tnext;
  tr->iter++;
  if ((e = set_timestep (tr))) return e;
  return advance_tracers (tr, tr->dt);
}

} // extern "C"
