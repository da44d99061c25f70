/* go_tree.c -- oracle: the reference's time step on a quadtree (2-D) or octree (3-D) with a
 * statically refined patch (coarse-fine stencils) in one periodic box: the case of
 * test/periodic/periodic.gfs with BOX = 1, 2, and its 3-D analogues (FTT_3D branches of the same
 * functions: interpolate_2D1, four children per face, FTT_CELLS = 8; the reference holds no 3-D
 * golden file: in 3-D this restatement is pinned by the 2-D files through the shared code, by the
 * uniform 3-D oracle on a uniform octree, and by the z-invariant case, tests/test_oracle_tree.py).
 * TEST INFRASTRUCTURE ONLY (see gfs_oracle.h): nothing of the product links or calls this.
 *
 * Pinned on the reference's test/periodic/r1.ref and r2.ref (tests/test_oracle_tree.py), and on
 * r0.ref through the uniform tree (BOX = 0), where it must also agree with go_timestep.c.
 *
 * Storage: level l of the tree is a dense (n+2)^2 array, n = 2^l, one ghost layer (the ghost
 * trees of the periodic GfsBoundary), index i + (n+2)*j, 1 <= i,j <= n inside, j grows with y;
 * flag[l][q] says whether cell q of level l exists and whether it is a leaf.  A cell is the pair
 * (level, index).  The reference's child numbering (ftt.c:301-316: bit 0 = +x, bit 1 = -y) and
 * traversal orders are reproduced by the traversal functions, not by the layout.
 *
 * What is restated, with the fine / coarse branches that go_timestep.c / go_poisson.c leave out:
 *   src/ftt.c:45-83,169-192,2013-2074   refinement with the neighbour and corner constraints
 *   src/ftt.c:689-926, ftt_internal.c   cell and face traversals
 *   src/fluid.c:64-93,178-197,283-309,364-396,434-475,778-893,2186-2198,2310-2324
 *   src/poisson.c:507-557,634-678,756-901,998-1269
 *   src/advection.c:27-99,132-180,267-343,398-435,513-587
 *   src/timestep.c:36-187,356-444,498-530,560-596,644-717,872-921,976-1016
 *   src/domain.c:2239-2288,2824-2923, src/simulation.c:432-557,1569-1633
 * Compile with -ffp-contract=off (oracle/Makefile). */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <assert.h>
#include "gfs_oracle.h"

#define GT_MAXL 12
#define OPP(d) ((d) ^ 1)
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#define G_MAXINT 2147483647
#ifndef M_PI
#define M_PI 3.14159265358979323846   /* math.h, as used by the reference */
#endif

enum { GT_NONE = 0, GT_LEAF = 1, GT_NODE = 2 };
#define GT_MAXTRACERS 2

typedef struct { int l, q; } Cell;            /* q < 0: no such cell (NULL) */
typedef struct { double * lev[GT_MAXL + 1]; } Var;
typedef struct { double a, b; } GfsGradient;  /* fluid.h: v = a*v(cell) + b */
typedef struct { double a, b, c; } Gradient;  /* fluid.c:55-59 */

typedef struct GtSim {
  int dim;                                    /* 2 or 3 */
  int nd, nc;                                 /* FTT_NEIGHBORS = 2 dim, FTT_CELLS = 2^dim */
  int depth;                                  /* deepest level present */
  int n[GT_MAXL + 1], r[GT_MAXL + 1];
  size_t size[GT_MAXL + 1];
  unsigned char * flag[GT_MAXL + 1];
  Var p, pmac, u[3], g[3], gmac[3];
  Var un[6], fv[6];                           /* GFS_STATE (cell)->f[d].un, f[d].v (advection)   */
  Var w[6];                                   /* GFS_STATE (cell)->f[d].v as Poisson weights     */
  GoMultilevelParams projection_params, approx_projection_params;
  double cfl, dt;                             /* GfsAdvectionParams */
  double t, end, tnext;
  unsigned i;
  /* sides with a GfsBoundary instead of the periodic image (GfsPoisson runs only): the condition of
     P on each side (GO_BC_*) and, per ghost cell, the value of its GfsFunction at the face centre */
  int side[6];                                /* GO_SIDE_PERIODIC / GO_SIDE_BOUNDARY */
  int bc_p[6];
  Var bcval;
  const Var * bc_var;                         /* the variable the conditions belong to (P) */
  int bc_homogeneous;                         /* set around the homogeneous BCs of a relax loop */
  /* GfsVariableTracer (variable.c:427-431): advected with the MAC velocities at the end of a step
     (gfs_advance_tracers, simulation.c:405-430); gradient 0 gfs_center_gradient, 1 van Leer (default) */
  Var tracer[GT_MAXTRACERS];
  int ntracers, tracer_gradient[GT_MAXTRACERS];
  /* conditions of U, V, W on GfsBoundary sides: GO_BC_SYMMETRY (the default GfsBc), _DIRICHLET, _NEUMANN
     with, per ghost cell, the value of the GfsFunction at the face centre (boundary.c:45-62,253-279,
     336-347 and their face_* forms) */
  int bc_u[3][6];
  Var bcu[3];
  int bcu_alloc;
  const Var * bc_owner;                       /* the variable whose (homogeneous) conditions a relax loop applies to dp */
  /* GfsSourceDiffusion {} U|V|W nu: constant implicit viscosity (source.c:933-1160) */
  double visc[3];
  GoMultilevelParams diffusion_params[3];
  double src[3];                              /* GfsSource {} U|V|W g: constant intensity (source.c:362-500) */
} GtSim;

static const Cell NOCELL = { 0, -1 };

/* ---- topology ----------------------------------------------------------------------------- */

static inline int cell_i (const GtSim * s, Cell c) { return c.q % s->r[c.l]; }
static inline int cell_j (const GtSim * s, Cell c) { return (c.q / s->r[c.l]) % s->r[c.l]; }
static inline int cell_k (const GtSim * s, Cell c) { return s->dim == 3 ? c.q / (s->r[c.l]*s->r[c.l]) : 1; }
static inline int exists (Cell c) { return c.q >= 0; }
static inline int is_leaf (const GtSim * s, Cell c) { return s->flag[c.l][c.q] == GT_LEAF; }
static inline double cell_size (Cell c) { return 1./(1 << c.l); }     /* ftt_cell_size, L = 1 */
static inline double * val (const Var * v, Cell c) { return &v->lev[c.l][c.q]; }

static Cell mkcell (const GtSim * s, int l, int i, int j, int k)
{
  Cell c = { l, -1 };
  if (l < 0 || l > s->depth || !s->flag[l] || i < 0 || j < 0 || i > s->n[l] + 1 || j > s->n[l] + 1)
    return c;
  if (s->dim == 3 && (k < 0 || k > s->n[l] + 1))
    return c;
  int q = i + s->r[l]*(j + (s->dim == 3 ? s->r[l]*k : 0));
  if (s->flag[l][q] != GT_NONE)
    c.q = q;
  return c;
}

/* FTT_CELL_ID: position among the siblings, ftt.c:301-316 (bit 0: +x, bit 1: -y, bit 2: -z) */
static inline int cell_id (const GtSim * s, Cell c)
{
  int i = cell_i (s, c), j = cell_j (s, c);
  return ((i + 1) & 1) + 2*(j & 1) + (s->dim == 3 ? 4*(cell_k (s, c) & 1) : 0);
}

/* ftt_cell_neighbor, ftt.h:518-573: the neighbour at the same level or, failing that, the
   (leaf) cell one level up that covers its place */
static Cell neighbor (const GtSim * s, Cell c, int d)
{
  static const int di[6] = { 1, -1, 0, 0, 0, 0 }, dj[6] = { 0, 0, 1, -1, 0, 0 }, dk[6] = { 0, 0, 0, 0, 1, -1 };
  int i = cell_i (s, c) + di[d], j = cell_j (s, c) + dj[d], k = cell_k (s, c) + dk[d];
  if (i < 0 || j < 0 || k < 0 || i > s->n[c.l] + 1 || j > s->n[c.l] + 1 || k > s->n[c.l] + 1)
    return NOCELL;
  Cell nb = mkcell (s, c.l, i, j, k);
  if (exists (nb) || c.l == 0)
    return nb;
  return mkcell (s, c.l - 1, (i + 1)/2, (j + 1)/2, (k + 1)/2);
}

static Cell child (const GtSim * s, Cell c, int k)
{
  return mkcell (s, c.l + 1, 2*cell_i (s, c) - 1 + (k & 1), 2*cell_j (s, c) - ((k >> 1) & 1),
		 2*cell_k (s, c) - ((k >> 2) & 1));
}

/* ftt_cell_children_direction, ftt.h:321-355: 2 children in 2-D, 4 in 3-D */
static int children_direction (const GtSim * s, Cell c, int d, Cell ch[4])
{
  static const int index2[4][2] = { {1, 3}, {0, 2}, {0, 1}, {2, 3} };
  static const int index3[6][4] = { {1, 3, 5, 7}, {0, 2, 4, 6}, {0, 1, 4, 5}, {2, 3, 6, 7},
				    {0, 1, 2, 3}, {4, 5, 6, 7} };
  int n = s->nc/2;
  for (int i = 0; i < n; i++)
    ch[i] = child (s, c, s->dim == 3 ? index3[d][i] : index2[d][i]);
  return n;
}

/* ftt_cell_child_corner, ftt.h:366-425: the child in the corner of the given directions (one per
   axis; an axis that is not given: d < 0) */
static Cell child_corner (const GtSim * s, Cell c, int d0, int d1, int d2)
{
  int d[3] = { d0, d1, d2 }, id = 0;
  for (int a = 0; a < 3; a++) {
    if (d[a] == 0) id |= 1;
    if (d[a] == 3) id |= 2;
    if (d[a] == 5) id |= 4;
  }
  return child (s, c, id);
}

/* fluid.c:200-213 (and advection.c:289-304): the directions, seen from the coarse neighbour, in
   which the fine cell sits */
static const int perpendicular2[4][4] =
  {{-1,  2, -1,  3},
   { 2, -1,  3, -1},
   { 1,  0, -1, -1},
   {-1, -1,  1,  0}};
static const int perpendicular3[6][8][2] =
  {{{-1,-1},{2,4},{-1,-1},{3,4},{-1,-1},{2,5},{-1,-1},{3,5}},
   {{2,4},{-1,-1},{3,4},{-1,-1},{2,5},{-1,-1},{3,5},{-1,-1}},
   {{4,1},{4,0},{-1,-1},{-1,-1},{5,1},{5,0},{-1,-1},{-1,-1}},
   {{-1,-1},{-1,-1},{4,1},{4,0},{-1,-1},{-1,-1},{5,1},{5,0}},
   {{1,2},{0,2},{1,3},{0,3},{-1,-1},{-1,-1},{-1,-1},{-1,-1}},
   {{-1,-1},{-1,-1},{-1,-1},{-1,-1},{1,2},{0,2},{1,3},{0,3}}};

/* ---- traversals, ftt.c:689-926 -------------------------------------------------------------- */

enum { T_ALL, T_LEAFS, T_NON_LEAFS, T_LEVEL, T_LEVEL_LEAFS, T_LEVEL_NON_LEAFS };
typedef void (* CellFunc) (GtSim * s, Cell c, void * data);

static void traverse_rec (GtSim * s, Cell c, int post, int flags, int max_depth,
			  CellFunc fn, void * data)
{
  int leaf = is_leaf (s, c), visit = 0, descend = !leaf;
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
  else { /* T_LEVEL_NON_LEAFS */
    visit = c.l == max_depth && !leaf;
    descend = !visit && !leaf;
  }
  if (visit && !post)
    (* fn) (s, c, data);
  if (descend)
    for (int k = 0; k < s->nc; k++) {
      Cell ch = child (s, c, k);
      if (exists (ch))
	traverse_rec (s, ch, post, flags, max_depth, fn, data);
    }
  if (visit && post)
    (* fn) (s, c, data);
}

static void cell_traverse (GtSim * s, int post, int flags, int max_depth, CellFunc fn, void * data)
{
  Cell root = mkcell (s, 0, 1, 1, 1);
  traverse_rec (s, root, post, flags, max_depth, fn, data);
}

/* ftt_cell_traverse_boundary (ftt.c:1153-1243): the cells of the traversal that touch side d */
typedef struct { int d; CellFunc fn; void * data; } BoundaryPar;

static void boundary_filter (GtSim * s, Cell c, void * data)
{
  BoundaryPar * b = data;
  int i = cell_i (s, c), j = cell_j (s, c), k = cell_k (s, c), n = s->n[c.l];
  if ((b->d == 0 && i == n) || (b->d == 1 && i == 1) || (b->d == 2 && j == n) || (b->d == 3 && j == 1) ||
      (b->d == 4 && k == n) || (b->d == 5 && k == 1))
    (* b->fn) (s, c, b->data);
}

/* a face as seen from `cell' (FttCellFace) */
typedef struct { Cell cell, neighbor; int d; } Face;
typedef void (* FaceFunc) (GtSim * s, const Face * f, void * data);
typedef struct { int d; FaceFunc fn; void * data; } FacePar;

/* traverse_face, ftt_internal.c:1-42, flags = FTT_TRAVERSE_LEAFS, max_depth = -1.  (The
   FTT_FLAG_TRAVERSED check only matters in the second pass, whose neighbours are ghost cells --
   never traversed -- so it is not kept.) */
static void traverse_face (GtSim * s, Cell cell, void * data)
{
  FacePar * p = data;
  Face face = { cell, neighbor (s, cell, p->d), p->d };
  if (!exists (face.neighbor))
    return;
  if (is_leaf (s, cell) && !is_leaf (s, face.neighbor)) {
    /* coarse -> fine */
    Cell ch[4];
    face.d = OPP (face.d);
    int n = children_direction (s, face.neighbor, face.d, ch);
    face.neighbor = face.cell;
    for (int i = 0; i < n; i++)
      if (exists (face.cell = ch[i]))
	(* p->fn) (s, &face, p->data);
  }
  else
    (* p->fn) (s, &face, p->data);
}

static void traverse_all_direct_faces (GtSim * s, Cell cell, void * data)
{
  FacePar * p = data;
  for (p->d = 0; p->d < s->nd; p->d += 2)
    traverse_face (s, cell, p);
}

/* ftt_face_traverse (ftt.c:2152-2215) through gfs_domain_face_traverse (domain.c:1725-1793):
   c < 0: FTT_XYZ; c = 0, 1: one component; leaves only */
static void face_traverse (GtSim * s, int c, FaceFunc fn, void * data)
{
  FacePar p = { 0, fn, data };
  if (c < 0) {
    cell_traverse (s, 0, T_LEAFS, -1, traverse_all_direct_faces, &p);
    for (int d = 1; d < s->nd; d += 2) {
      BoundaryPar b = { d, traverse_face, &p };
      p.d = d;
      cell_traverse (s, 0, T_LEAFS, -1, boundary_filter, &b);
    }
  }
  else {
    p.d = 2*c;
    cell_traverse (s, 0, T_LEAFS, -1, traverse_face, &p);
    BoundaryPar b = { 2*c + 1, traverse_face, &p };
    p.d = 2*c + 1;
    cell_traverse (s, 0, T_LEAFS, -1, boundary_filter, &b);
  }
}

static inline int fine_coarse (const Face * f) { return f->neighbor.l < f->cell.l; } /* ftt_face_type */

/* ---- variables and boundary conditions ----------------------------------------------------- */

static void var_alloc (GtSim * s, Var * v)
{
  for (int l = 0; l <= s->depth; l++)
    v->lev[l] = calloc (s->size[l], sizeof (double));
}

static void var_free (GtSim * s, Var * v)
{
  for (int l = 0; l <= s->depth; l++)
    free (v->lev[l]);
}

/* gfs_domain_copy_bc / gfs_domain_bc (domain.c:846-920) on a box whose four sides are periodic
   (boundary.c:1240-1451): the ghost cells selected by (flags, max_depth) take the value of their
   periodic image.  v1 == v for a plain BC; the homogeneous BC of a periodic side is the same copy. */
/* the ghost cells of side `side' of level l and their periodic images: calls fn (G, image) */
typedef void (* GhostFunc) (GtSim * s, int l, int side, int G, int image, void * data);
static void ghost_traverse (GtSim * s, int l, GhostFunc fn, void * data)
{
  int n = s->n[l], r = s->r[l];
  for (int side = 0; side < s->nd; side++)
    for (int tb = 1; tb <= (s->dim == 3 ? n : 1); tb++)
      for (int ta = 1; ta <= n; ta++) {
	int g[3], im[3], a = side/2;     /* axis of the side; the two others run over 1..n */
	int o1 = a == 0 ? 1 : 0, o2 = a == 2 ? 1 : 2;
	g[a] = (side & 1) ? 0 : n + 1;
	im[a] = (side & 1) ? n : 1;
	g[o1] = im[o1] = ta;
	g[o2] = im[o2] = s->dim == 3 ? tb : 0;
	int G = g[0] + r*(g[1] + r*g[2]), I = im[0] + r*(im[1] + r*im[2]);
	(* fn) (s, l, side, G, I, data);
      }
}

typedef struct { Var * v; int flags, max_depth; } BcPar;
/* the component of a vector a variable is (gfs_variable_set_vector: U, V, W, g, gmac), or -1 */
static int var_component (const GtSim * s, const Var * v)
{
  for (int c = 0; c < 3; c++)
    if (v == &s->u[c] || v == &s->g[c] || v == &s->gmac[c])
      return c;
  return -1;
}

/* the interior cell a ghost cell of side `side' touches */
static int ghost_own (const GtSim * s, int l, int side, int G)
{
  int r = s->r[l], a = side/2;
  int stride = a == 0 ? 1 : a == 1 ? r : r*r;
  return (side & 1) ? G + stride : G - stride;
}

static void bc_ghost (GtSim * s, int l, int side, int G, int image, void * data)
{
  BcPar * p = data;
  unsigned char f = s->flag[l][G];
  if (f == GT_NONE)
    return;
  int take = p->flags == T_LEAFS ? f == GT_LEAF :
    p->flags == T_LEVEL_LEAFS ? (l == p->max_depth || f == GT_LEAF) : 1;
  if (!take)
    return;
  if (s->side[side] == GO_SIDE_PERIODIC) {
    p->v->lev[l][G] = p->v->lev[l][image];
    return;
  }
  /* GfsBoundary: symmetry (default, scalar) boundary.c:45-62, Dirichlet :253-279, Neumann :336-347 */
  double nb = p->v->lev[l][ghost_own (s, l, side, G)];
  double h = 1./s->n[l];
  const Var * owner = s->bc_owner ? s->bc_owner : p->v;     /* gfs_domain_homogeneous_bc (ov, v) */
  int comp = var_component (s, owner);
  int ucomp = -1;
  for (int c = 0; c < 3; c++)
    if (owner == &s->u[c]) ucomp = c;
  if (ucomp >= 0 && s->bc_u[ucomp][side] != GO_BC_SYMMETRY) {
    int kind = s->bc_u[ucomp][side];
    if (s->bc_homogeneous)
      p->v->lev[l][G] = kind == GO_BC_DIRICHLET ? - nb : nb;
    else {
      double value = s->bcu[ucomp].lev[l][G];
      p->v->lev[l][G] = kind == GO_BC_DIRICHLET ? 2.*value - nb : nb + value*h;
    }
    return;
  }
  if (comp >= 0) {      /* symmetry (the default GfsBc) of a vector component, boundary.c:45-62 */
    p->v->lev[l][G] = comp == side/2 ? - nb : nb;
    return;
  }
  for (int t = 0; t < s->ntracers; t++)
    if (owner == &s->tracer[t]) {     /* a scalar with the default GfsBc: symmetry, boundary.c:45-62 */
      p->v->lev[l][G] = nb;
      return;
    }
  int kind = s->bc_p[side];
  if (s->bc_homogeneous)
    p->v->lev[l][G] = kind == GO_BC_DIRICHLET ? - nb : nb;
  else {
    double value = s->bcval.lev[l][G];
    p->v->lev[l][G] = kind == GO_BC_DIRICHLET ? 2.*value - nb : kind == GO_BC_NEUMANN ? nb + value*h : nb;
  }
}

static void bc (GtSim * s, Var * v, int flags, int max_depth)
{
  BcPar p = { v, flags, max_depth };
  for (int l = 0; l <= s->depth; l++) {
    if (max_depth >= 0 && l > max_depth)
      break;
    ghost_traverse (s, l, bc_ghost, &p);
  }
}

/* ---- tree construction --------------------------------------------------------------------- */

/* oct_new, ftt.c:45-83 with check_neighbors: a cell about to get children first makes sure that
   none of its neighbours is coarser than itself */
static void refine_single (GtSim * s, int l, int i, int j, int k)
{
  assert (l < GT_MAXL);
  int r = s->r[l], q = i + r*(j + (s->dim == 3 ? r*k : 0));
  assert (s->flag[l][q] == GT_LEAF);
  static const int di[6] = { 1, -1, 0, 0, 0, 0 }, dj[6] = { 0, 0, 1, -1, 0, 0 }, dk[6] = { 0, 0, 0, 0, 1, -1 };
  for (int d = 0; d < s->nd; d++) {
    int ni = i + di[d], nj = j + dj[d], nk = k + dk[d];
    if (ni < 1 || nj < 1 || ni > s->n[l] || nj > s->n[l] || (s->dim == 3 && (nk < 1 || nk > s->n[l])))
      continue; /* the ghost trees are matched at the end (gfs_domain_match) */
    if (s->flag[l][ni + r*(nj + (s->dim == 3 ? r*nk : 0))] == GT_NONE) {
      int pi = (ni + 1)/2, pj = (nj + 1)/2, pk = (nk + 1)/2, pr = s->r[l - 1];
      if (s->flag[l - 1][pi + pr*(pj + (s->dim == 3 ? pr*pk : 0))] == GT_LEAF)
	refine_single (s, l - 1, pi, pj, pk);
    }
  }
  s->flag[l][q] = GT_NODE;
  if (!s->flag[l + 1])
    s->flag[l + 1] = calloc (s->size[l + 1], 1);
  int cr = s->r[l + 1];
  for (int c = 0; c < s->nc; c++)
    s->flag[l + 1][2*i - 1 + (c & 1) + cr*(2*j - ((c >> 1) & 1) + (s->dim == 3 ? cr*(2*k - ((c >> 2) & 1)) : 0))] = GT_LEAF;
}

/* ftt_cell_refine, ftt.c:169-192, with refine_maxlevel, refine.c:35-38: `refine' is the GfsFunction
   of the GfsRefine object evaluated at the centre of the cell */
typedef double (* GtRefineFunc) (double x, double y, double z, void * ctx);

static void refine_rec (GtSim * s, int l, int i, int j, int k, GtRefineFunc refine, void * ctx)
{
  int r = s->r[l], q = i + r*(j + (s->dim == 3 ? r*k : 0));
  if (s->flag[l][q] == GT_LEAF) {
    double h = 1./s->n[l];
    double x = -0.5 + (i - 0.5)*h, y = -0.5 + (j - 0.5)*h, z = s->dim == 3 ? -0.5 + (k - 0.5)*h : 0.;
    if (!(l < (* refine) (x, y, z, ctx)))
      return;
    refine_single (s, l, i, j, k);
  }
  for (int c = 0; c < s->nc; c++)
    refine_rec (s, l + 1, 2*i - 1 + (c & 1), 2*j - ((c >> 1) & 1), 2*k - ((c >> 2) & 1), refine, ctx);
}

/* ftt_refine_corner, ftt.c:2013-2074 */
static int refine_corner (const GtSim * s, Cell cell)
{
  static const int perp2[4][2] = { {2, 3}, {2, 3}, {1, 0}, {1, 0} };
  static const int perp3[6][4][2] =
    {{{4,2},{4,3},{5,2},{5,3}},
     {{4,2},{4,3},{5,2},{5,3}},
     {{4,1},{4,0},{5,1},{5,0}},
     {{4,1},{4,0},{5,1},{5,0}},
     {{2,1},{2,0},{3,1},{3,0}},
     {{2,1},{2,0},{3,1},{3,0}}};
  for (int i = 0; i < s->nd; i++) {
    Cell n = neighbor (s, cell, i);
    if (exists (n) && !is_leaf (s, n)) {
      Cell ch[4];
      int k = children_direction (s, n, OPP (i), ch);
      for (int j = 0; j < k; j++)
	if (exists (ch[j])) {
	  if (s->dim == 2) {
	    Cell nc = neighbor (s, ch[j], perp2[i][j]);
	    if (exists (nc) && !is_leaf (s, nc))
	      return 1;
	  }
	  else {
	    Cell nc0 = neighbor (s, ch[j], perp3[i][j][0]);
	    if (exists (nc0) && !is_leaf (s, nc0))
	      return 1;
	    Cell nc1 = neighbor (s, ch[j], perp3[i][j][1]);
	    if (exists (nc1) && !is_leaf (s, nc1))
	      return 1;
	  }
	  if (!is_leaf (s, ch[j]))
	    return 1;
	}
    }
  }
  return 0;
}

static void refine_cell_corner (GtSim * s, Cell c, void * data) /* simulation.c:1105-1109 */
{
  if (is_leaf (s, c) && refine_corner (s, c))
    refine_single (s, c.l, cell_i (s, c), cell_j (s, c), cell_k (s, c));
}

typedef struct { int bad; } MatchPar;
static void match_ghost (GtSim * s, int l, int side, int G, int image, void * data)
{
  MatchPar * m = data;
  /* the cell along the side itself: the image seen from the opposite side */
  int n = s->n[l], r = s->r[l], a = side/2;
  int stride = a == 0 ? 1 : a == 1 ? r : r*r;
  int own = (side & 1) ? G + stride : G - stride;
  if (s->side[side] != GO_SIDE_PERIODIC) {    /* the ghost tree of a GfsBoundary matches its own side */
    s->flag[l][G] = s->flag[l][own];
    return;
  }
  if (s->flag[l][own] != s->flag[l][image])
    m->bad = 1;
  s->flag[l][G] = s->flag[l][image];
  (void) n;
}

static void build_tree (GtSim * s, GtRefineFunc refine, void * ctx)
{
  for (int l = 0; l <= GT_MAXL; l++) {
    s->n[l] = 1 << l;
    s->r[l] = s->n[l] + 2;
    s->size[l] = (size_t) s->r[l]*s->r[l]*(s->dim == 3 ? s->r[l] : 1);
  }
  s->flag[0] = calloc (s->size[0], 1);
  s->depth = GT_MAXL; /* while the tree grows: every level refine_single allocates is valid */
  s->flag[0][1 + s->r[0]*(1 + (s->dim == 3 ? s->r[0] : 0))] = GT_LEAF;
  refine_rec (s, 0, 1, 1, 1, refine, ctx);
  /* gfs_domain_depth */
  int depth = 0;
  for (int l = 0; l <= GT_MAXL && s->flag[l]; l++)
    depth = l;
  /* gfs_simulation_refine, simulation.c:1226-1231 */
  for (int l = depth - 2; l >= 0; l--)
    cell_traverse (s, 0, T_LEVEL, l, refine_cell_corner, NULL);
  s->depth = depth;
  /* gfs_domain_match: the ghost trees of the periodic sides mirror the cells they face; the
     refined patches of the cases restated here stay away from the sides (both cells of a periodic
     pair at the same refinement) */
  MatchPar m = { 0 };
  for (int l = 0; l <= depth; l++)
    ghost_traverse (s, l, match_ghost, &m);
  assert (!m.bad);
}

/* ---- fluid.c: neighbour values and gradients ---------------------------------------------- */

/* average_neighbor_value, fluid.c:64-93 */
static double average_neighbor_value (const GtSim * s, const Face * face, const Var * v, double * x)
{
  assert (face->neighbor.l == face->cell.l);
  if (is_leaf (s, face->neighbor))
    return *val (v, face->neighbor);
  Cell ch[4];
  double av = 0., a = 0.;
  int n = children_direction (s, face->neighbor, OPP (face->d), ch);
  for (int i = 0; i < n; i++)
    if (exists (ch[i])) {
      double w = 1.;
      a += w;
      av += w*(*val (v, ch[i]));
    }
  if (a > 0.) {
    *x = 3./4.;
    return av/a;
  }
  return *val (v, face->cell);
}

/* interpolate_1D1, fluid.c:178-197 */
static GfsGradient interpolate_1D1 (const GtSim * s, Cell cell, int d, double x, const Var * v)
{
  GfsGradient p = { 1., 0. };
  Face f = { cell, neighbor (s, cell, d), d };
  if (exists (f.neighbor)) {
    double x2 = 1.;
    double p2 = average_neighbor_value (s, &f, v, &x2);
    double a2 = x/x2;
    p.b += a2*p2;
    p.a -= a2;
  }
  return p;
}

/* interpolate_2D1, fluid.c:214-245 (3-D) */
static GfsGradient interpolate_2D1 (const GtSim * s, Cell cell, int d1, int d2, double x, double y,
				    const Var * v)
{
  GfsGradient p = { 1., 0. };
  Face f1 = { cell, neighbor (s, cell, d1), d1 };
  if (exists (f1.neighbor)) {
    double y1 = 1.;
    double p1 = average_neighbor_value (s, &f1, v, &y1);
    double a1 = y/y1;
    p.b += a1*p1;
    p.a -= a1;
  }
  Face f2 = { cell, neighbor (s, cell, d2), d2 };
  if (exists (f2.neighbor)) {
    double x2 = 1.;
    double p2 = average_neighbor_value (s, &f2, v, &x2);
    double a2 = x/x2;
    p.b += a2*p2;
    p.a -= a2;
  }
  return p;
}

/* the interpolation in the coarse neighbour of a fine-coarse face towards the fine cell:
   interpolate_1D1 (FTT_2D) or interpolate_2D1, as gradient_fine_coarse and gfs_neighbor_value call it */
static GfsGradient interpolate_coarse (const GtSim * s, const Face * face, const Var * v)
{
  int id = cell_id (s, face->cell);
  if (s->dim == 2) {
    int dp = perpendicular2[face->d][id];
    assert (dp >= 0);
    return interpolate_1D1 (s, face->neighbor, dp, 1./4., v);
  }
  const int * dp = perpendicular3[face->d][id];
  assert (dp[0] >= 0 && dp[1] >= 0);
  return interpolate_2D1 (s, face->neighbor, dp[0], dp[1], 1./4., 1./4., v);
}

/* gradient_fine_coarse, fluid.c:283-309 */
static Gradient gradient_fine_coarse (const GtSim * s, const Face * face, const Var * v)
{
  Gradient g;
  assert (fine_coarse (face));
  GfsGradient p = interpolate_coarse (s, face, v);
  g.a = 2./3.;
  g.b = 2.*p.a/3.;
  g.c = 2.*p.b/3.;
  return g;
}

/* gfs_neighbor_value, fluid.c:364-396 */
static double neighbor_value (const GtSim * s, const Face * face, const Var * v, double * x)
{
  if (face->neighbor.l == face->cell.l)
    return average_neighbor_value (s, face, v, x);
  GfsGradient vc = interpolate_coarse (s, face, v);
  *x = 3./2.;
  return vc.a*(*val (v, face->neighbor)) + vc.b;
}

/* gfs_center_gradient, fluid.c:434-475 */
static double center_gradient (const GtSim * s, Cell cell, int c, const Var * v)
{
  int d = 2*c;
  Face f1 = { cell, neighbor (s, cell, OPP (d)), OPP (d) };
  double v0 = *val (v, cell);
  if (exists (f1.neighbor)) {
    Face f2 = { cell, neighbor (s, cell, d), d };
    double x1 = 1., v1;
    v1 = neighbor_value (s, &f1, v, &x1);
    if (exists (f2.neighbor)) {
      double x2 = 1., v2;
      v2 = neighbor_value (s, &f2, v, &x2);
      return (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
    }
    return (v0 - v1)/x1;
  }
  Face f2 = { cell, neighbor (s, cell, d), d };
  if (exists (f2.neighbor)) {
    double x2 = 1.;
    return (neighbor_value (s, &f2, v, &x2) - v0)/x2;
  }
  return 0.;
}

/* gfs_center_van_leer_gradient, fluid.c:522-561 */
static double center_van_leer_gradient (const GtSim * s, Cell cell, int c, const Var * v)
{
  int d = 2*c;
  Face f1 = { cell, neighbor (s, cell, OPP (d)), OPP (d) };
  if (exists (f1.neighbor)) {
    Face f2 = { cell, neighbor (s, cell, d), d };
    if (exists (f2.neighbor)) {
      double x1 = 1., x2 = 1., v0, v1, v2;
      v0 = *val (v, cell);
      v1 = neighbor_value (s, &f1, v, &x1);
      v2 = neighbor_value (s, &f2, v, &x2);
      double s1 = 2.*(v0 - v1);
      double s2 = 2.*(v2 - v0);
      if (s1*s2 <= 0.)
	return 0.;
      double s0 = (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
      if (fabs (s2) < fabs (s1))
	s1 = s2;
      if (fabs (s0) < fabs (s1))
	return s0;
      return s1;
    }
  }
  return 0.;
}

/* gfs_face_gradient, fluid.c:778-829 */
static void face_gradient (const GtSim * s, const Face * face, GfsGradient * g, const Var * v,
			   int max_level)
{
  g->a = g->b = 0.;
  if (!exists (face->neighbor))
    return;
  int level = face->cell.l;
  if (face->neighbor.l < level) {
    Gradient gcf = gradient_fine_coarse (s, face, v);
    g->a = gcf.a;
    g->b = gcf.b*(*val (v, face->neighbor)) + gcf.c;
  }
  else if (level == max_level || is_leaf (s, face->neighbor)) {
    g->a = 1.;
    g->b = *val (v, face->neighbor);
  }
  else {
    Cell ch[4];
    Face f;
    f.d = OPP (face->d);
    int n = children_direction (s, face->neighbor, f.d, ch);
    f.neighbor = face->cell;
    for (int i = 0; i < n; i++)
      if (exists (f.cell = ch[i])) {
	Gradient gcf = gradient_fine_coarse (s, &f, v);
	double sf = 1.;
	g->a += sf*gcf.b;
	g->b += sf*(gcf.a*(*val (v, f.cell)) - gcf.c);
      }
    double sf = 1.*n/2.;
    g->a /= sf;
    g->b /= sf;
  }
}

/* face_weighted_gradient, fluid.c:833-893 (dimension = FTT_DIMENSION) */
static void face_weighted_gradient (const GtSim * s, const Face * face, GfsGradient * g,
				    const Var * v, int max_level)
{
  g->a = g->b = 0.;
  if (!exists (face->neighbor))
    return;
  int level = face->cell.l;
  if (face->neighbor.l < level) {
    double w = *val (&s->w[face->d], face->cell);
    Gradient gcf = gradient_fine_coarse (s, face, v);
    g->a = w*gcf.a;
    g->b = w*(gcf.b*(*val (v, face->neighbor)) + gcf.c);
  }
  else if (level == max_level || is_leaf (s, face->neighbor)) {
    double w = *val (&s->w[face->d], face->cell);
    g->a = w;
    g->b = w*(*val (v, face->neighbor));
  }
  else {
    Cell ch[4];
    Face f;
    f.d = OPP (face->d);
    int n = children_direction (s, face->neighbor, f.d, ch);
    f.neighbor = face->cell;
    for (int i = 0; i < n; i++)
      if (exists (f.cell = ch[i])) {
	double w = *val (&s->w[f.d], f.cell);
	Gradient gcf = gradient_fine_coarse (s, &f, v);
	g->a += w*gcf.b;
	g->b += w*(gcf.a*(*val (v, f.cell)) - gcf.c);
      }
    if (s->dim > 2) {
      g->a /= n/2.;
      g->b /= n/2.;
    }
  }
}

/* gfs_face_interpolated_value, fluid.c:2186-2198 */
static double face_interpolated_value (const GtSim * s, const Face * face, const Var * v)
{
  double x1 = 1., v1;
  if (exists (face->neighbor)) {
    assert (is_leaf (s, face->neighbor) || face->neighbor.l < face->cell.l);
    v1 = neighbor_value (s, face, v, &x1);
    return ((x1 - 0.5)*(*val (v, face->cell)) + 0.5*v1)/x1;
  }
  return *val (v, face->cell);
}

/* ---- Poisson: poisson.c ------------------------------------------------------------------- */

static void reset_coeff (GtSim * s, Cell c, void * data) /* poisson.c:756-766 */
{
  for (int d = 0; d < s->nd; d++)
    *val (&s->w[d], c) = 0.;
}

static void poisson_coeff (GtSim * s, const Face * face, void * data) /* poisson.c:768-799 */
{
  double alpha = 1.;
  double v = 1.*alpha*1./1.;
  *val (&s->w[face->d], face->cell) += v;
  if (!fine_coarse (face))
    *val (&s->w[OPP (face->d)], face->neighbor) += v;
  else
    *val (&s->w[OPP (face->d)], face->neighbor) += v/(s->nc/2); /* FTT_CELLS_DIRECTION */
}

static void face_coeff_from_below (GtSim * s, Cell cell, void * data) /* poisson.c:826-853 */
{
  unsigned neighbors = 0;
  for (int d = 0; d < s->nd; d++) {
    Cell ch[4];
    double * f = val (&s->w[d], cell);
    *f = 0.;
    int n = children_direction (s, cell, d, ch);
    for (int i = 0; i < n; i++)
      if (exists (ch[i]))
	*f += *val (&s->w[d], ch[i]);
    *f /= n;
    Cell nb = neighbor (s, cell, d);
    if (*f != 0. && exists (nb)) {
      int i = cell_i (s, nb), j = cell_j (s, nb), k = cell_k (s, nb), nn = s->n[nb.l];
      if (i >= 1 && i <= nn && j >= 1 && j <= nn && k >= 1 && k <= nn) /* !GFS_CELL_IS_BOUNDARY */
	neighbors++;
    }
  }
  if (neighbors == 1)
    for (int d = 0; d < s->nd; d++)
      *val (&s->w[d], cell) = 0.;
}

/* gfs_poisson_coefficients (alpha = NULL), poisson.c:855-901.  The ghost cells are reset with
   the cells they mirror (their weights are written by the faces of the sides, never read). */
static void poisson_coefficients (GtSim * s)
{
  for (int d = 0; d < s->nd; d++)
    for (int l = 0; l <= s->depth; l++)
      memset (s->w[d].lev[l], 0, s->size[l]*sizeof (double));
  cell_traverse (s, 0, T_ALL, -1, reset_coeff, NULL);
  face_traverse (s, -1, poisson_coeff, NULL);
  cell_traverse (s, 1, T_NON_LEAFS, -1, face_coeff_from_below, NULL);
}

typedef struct { Var * u, * rhs, * dia, * res; int maxlevel; double omega; } RelaxParams;

static void relax2D (GtSim * s, Cell cell, void * data) /* poisson.c:532-557 */
{
  RelaxParams * p = data;
  GfsGradient g, ng;
  g.a = *val (p->dia, cell);
  g.b = 0.;
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < s->nd; f.d++) {
    f.neighbor = neighbor (s, cell, f.d);
    if (exists (f.neighbor)) {
      face_weighted_gradient (s, &f, &ng, p->u, p->maxlevel);
      g.a += ng.a;
      g.b += ng.b;
    }
  }
  if (g.a != 0.) {
    if (s->dim == 2)        /* relax2D, poisson.c:532-557 */
      *val (p->u, cell) = (1. - p->omega)*(*val (p->u, cell))
	+ p->omega*(g.b - *val (p->rhs, cell))/g.a;
    else                    /* relax, poisson.c:507-530 */
      *val (p->u, cell) = (g.b - *val (p->rhs, cell))/g.a;
  }
  else
    *val (p->u, cell) = 0.;
}

static void residual_set2D (GtSim * s, Cell cell, void * data) /* poisson.c:657-678 */
{
  RelaxParams * p = data;
  GfsGradient g, ng;
  g.a = *val (p->dia, cell);
  g.b = 0.;
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < s->nd; f.d++) {
    f.neighbor = neighbor (s, cell, f.d);
    if (exists (f.neighbor)) {
      face_weighted_gradient (s, &f, &ng, p->u, p->maxlevel);
      g.a += ng.a;
      g.b += ng.b;
    }
  }
  *val (p->res, cell) = *val (p->rhs, cell) - (g.b - *val (p->u, cell)*g.a);
}

/* gfs_residual on the leaves, poisson.c:721-747 */
static void residual (GtSim * s, Var * u, Var * rhs, Var * dia, Var * res)
{
  RelaxParams p = { u, rhs, dia, res, -1, 1. };
  cell_traverse (s, 0, T_LEAFS, -1, residual_set2D, &p);
}

typedef struct { Var * res; double bias; GoNorm n; } ResData;

static void norm_add (GoNorm * n, double v, double weight) /* gfs_norm_add, fluid.c:2139-2151 */
{
  n->bias += weight*v;
  v = fabs (v);
  if (weight != 0. && v > n->infty)
    n->infty = v;
  n->first += weight*v;
  n->second += weight*v*v;
  n->w += weight;
}

static void norm_update (GoNorm * n) /* gfs_norm_update, fluid.c:2159-2171 */
{
  if (n->w > 0.0) {
    n->bias /= n->w;
    n->first /= n->w;
    n->second = sqrt (n->second/n->w);
  }
  else
    n->infty = 0.0;
}

static void add_norm_residual (GtSim * s, Cell cell, void * data) /* domain.c:2239-2246 */
{
  ResData * p = data;
  double size = cell_size (cell);
  norm_add (&p->n, *val (p->res, cell)/(1.*size*size), 1.);
  p->bias += *val (p->res, cell);
}

static GoNorm norm_residual (GtSim * s, double dt, Var * res) /* domain.c:2264-2288 */
{
  ResData p = { res, 0., { 0., 0., 0., - DBL_MAX, 0. } };
  cell_traverse (s, 0, T_LEAFS, -1, add_norm_residual, &p);
  norm_update (&p.n);
  dt *= dt;
  p.n.bias = p.bias*dt;
  p.n.first *= dt;
  p.n.second *= dt;
  p.n.infty *= dt;
  return p.n;
}

static void get_from_below_2D (GtSim * s, Cell cell, void * data) /* poisson.c:1057-1068 */
{
  Var * v = data;
  double sum = 0.;
  for (int k = 0; k < s->nc; k++) {
    Cell ch = child (s, cell, k);
    if (exists (ch))
      sum += *val (v, ch);
  }
  *val (v, cell) = s->dim == 2 ? sum : sum/2.;   /* get_from_below_2D / _3D, poisson.c:1044-1068 */
}

static void get_from_above (GtSim * s, Cell parent, void * data) /* poisson.c:1005-1042 */
{
  Var * v = data;
  int level = parent.l;
  double h[3];
  for (int c = 0; c < s->dim; c++) {
    Face f;
    GfsGradient g;
    f.cell = parent;
    f.d = 2*c;
    f.neighbor = neighbor (s, parent, f.d);
    face_gradient (s, &f, &g, v, level);
    double g1 = g.b - g.a*(*val (v, parent));
    f.d = 2*c + 1;
    f.neighbor = neighbor (s, parent, f.d);
    face_gradient (s, &f, &g, v, level);
    double g2 = g.b - g.a*(*val (v, parent));
    h[c] = (g1 - g2)/2.;
  }
  for (int k = 0; k < s->nc; k++) {
    Cell ch = child (s, parent, k);
    if (exists (ch)) {
      /* ftt_cell_relative_pos, ftt.c:327-340 */
      double px = (k & 1) ? 0.25 : -0.25, py = (k & 2) ? -0.25 : 0.25, pz = (k & 4) ? -0.25 : 0.25;
      *val (v, ch) = *val (v, parent);
      *val (v, ch) += px*h[0];
      *val (v, ch) += py*h[1];
      if (s->dim == 3)
	*val (v, ch) += pz*h[2];
    }
  }
}

static void cell_reset (GtSim * s, Cell c, void * data) { *val ((Var *) data, c) = 0.; }

/* relax_loop, poisson.c:1070-1089 (the homogeneous BC of a periodic side is the periodic copy) */
static void relax_loop (GtSim * s, Var * dp, RelaxParams * q, unsigned nrelax)
{
  s->bc_homogeneous = 1;     /* gfs_domain_homogeneous_bc */
  bc (s, dp, T_LEVEL_LEAFS, q->maxlevel);
  for (unsigned n = 0; n < nrelax - 1; n++) {
    cell_traverse (s, 0, T_LEVEL_LEAFS, q->maxlevel, relax2D, q);
    bc (s, dp, T_LEVEL_LEAFS, q->maxlevel);
  }
  s->bc_homogeneous = 0;
  cell_traverse (s, 0, T_LEVEL_LEAFS, q->maxlevel, relax2D, q);
}

typedef struct { Var * u, * dp; } CorrectData;
static void correct (GtSim * s, Cell c, void * data) /* poisson.c:998-1003 */
{
  CorrectData * d = data;
  *val (d->u, c) += *val (d->dp, c);
}

/* gfs_poisson_cycle, poisson.c:1105-1178 */
static void poisson_cycle (GtSim * s, GoMultilevelParams * p, Var * u, Var * rhs, Var * dia, Var * res)
{
  Var dp;
  var_alloc (s, &dp);
  unsigned minlevel = p->minlevel;
  cell_traverse (s, 1, T_NON_LEAFS, -1, get_from_below_2D, res);
  unsigned nrelax = p->nrelax;
  for (unsigned l = minlevel; l < p->depth; l++)
    nrelax *= p->erelax;
  RelaxParams q = { &dp, res, dia, NULL, (int) minlevel, p->omega };
  cell_traverse (s, 0, T_LEVEL_LEAFS, q.maxlevel, cell_reset, &dp);
  relax_loop (s, &dp, &q, nrelax);
  nrelax /= p->erelax;
  for (q.maxlevel = minlevel + 1; q.maxlevel <= (int) p->depth; q.maxlevel++, nrelax /= p->erelax) {
    cell_traverse (s, 0, T_LEVEL_NON_LEAFS, q.maxlevel - 1, get_from_above, &dp);
    relax_loop (s, &dp, &q, nrelax);
  }
  CorrectData cd = { u, &dp };
  cell_traverse (s, 0, T_LEAFS, -1, correct, &cd);
  bc (s, u, T_LEAFS, -1);
  residual (s, u, rhs, dia, res);
  var_free (s, &dp);
}

/* gfs_poisson_solve, poisson.c:1225-1269 */
static void poisson_solve (GtSim * s, GoMultilevelParams * par, Var * lhs, Var * rhs, Var * res,
			   Var * dia, double dt)
{
  unsigned minlevel = par->minlevel;
  par->depth = s->depth;
  par->niter = 0;
  residual (s, lhs, rhs, dia, res);
  par->residual_before = par->residual = norm_residual (s, dt, res);
  double res_max_before = par->residual.infty;
  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    poisson_cycle (s, par, lhs, rhs, dia, res);
    par->residual = norm_residual (s, dt, res);
    if (par->residual.infty == res_max_before)
      break;
    if (par->residual.infty > res_max_before/1.1 && par->minlevel < par->depth)
      par->minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }
  par->minlevel = minlevel;
}

/* ---- projection: timestep.c --------------------------------------------------------------- */

/* ---- implicit diffusion: poisson.c:1271-1690, timestep.c:735-788,923-949 (constant D, alpha = NULL) - */

static void get_from_below_intensive (GtSim * s, Cell cell, void * data);

typedef struct { double v; } DiffCoef;
static void diffusion_coef (GtSim * s, const Face * face, void * data) /* poisson.c:1280-1303 */
{
  double v = ((DiffCoef *) data)->v;
  *val (&s->w[face->d], face->cell) = v;
  if (!fine_coarse (face))
    *val (&s->w[OPP (face->d)], face->neighbor) = v;
  else
    *val (&s->w[OPP (face->d)], face->neighbor) += v/(s->nc/2); /* FTT_CELLS_DIRECTION */
}

static void diffusion_mixed_coeff (GtSim * s, Cell c, void * data) /* poisson.c:1305-1334 */
{
  reset_coeff (s, c, NULL);
  *val ((Var *) data, c) = 1.*1.;      /* rhoc = rho*fraction */
}

/* gfs_diffusion_coefficients, poisson.c:1350-1390 */
static void diffusion_coefficients (GtSim * s, double D, double dt, double beta, Var * rhoc)
{
  for (int d = 0; d < s->nd; d++)
    for (int l = 0; l <= s->depth; l++)
      memset (s->w[d].lev[l], 0, s->size[l]*sizeof (double));
  cell_traverse (s, 0, T_ALL, -1, diffusion_mixed_coeff, rhoc);
  DiffCoef c = { 1.*(beta*dt)*D*1./1. };
  face_traverse (s, -1, diffusion_coef, &c);
  cell_traverse (s, 1, T_NON_LEAFS, -1, face_coeff_from_below, NULL);
}

typedef struct { Var * u, * rhs, * dia, * res; int maxlevel; double beta; } DiffParams;

static void diffusion_rhs (GtSim * s, Cell cell, void * data) /* poisson.c:1392-1421 */
{
  DiffParams * p = data;
  double f = 0., h = cell_size (cell), value = *val (p->u, cell);
  Face face;
  face.cell = cell;
  for (face.d = 0; face.d < s->nd; face.d++) {
    GfsGradient g;
    face.neighbor = neighbor (s, cell, face.d);
    face_weighted_gradient (s, &face, &g, p->u, -1);   /* gfs_face_cm_weighted_gradient, cm = 1 */
    f += g.b - g.a*value;
  }
  *val (p->rhs, cell) += p->beta*f/(h*h*(*val (p->dia, cell)));
}

static void diffusion_relax (GtSim * s, Cell cell, void * data) /* poisson.c:1455-1484 */
{
  DiffParams * p = data;
  GfsGradient g = { 0., 0. };
  double h = cell_size (cell);
  Face face;
  face.cell = cell;
  for (face.d = 0; face.d < s->nd; face.d++) {
    GfsGradient ng;
    face.neighbor = neighbor (s, cell, face.d);
    face_weighted_gradient (s, &face, &ng, p->u, p->maxlevel);
    g.a += ng.a;
    g.b += ng.b;
  }
  double a = *val (p->dia, cell)*h*h;
  g.a = 1. + g.a/a;
  *val (p->u, cell) = (g.b/a + *val (p->res, cell))/g.a;
}

static void diffusion_residual_cell (GtSim * s, Cell cell, void * data) /* poisson.c:1519-1556 */
{
  DiffParams * p = data;
  GfsGradient g = { 0., 0. };
  double h = cell_size (cell), a = *val (p->dia, cell);
  Face face;
  face.cell = cell;
  for (face.d = 0; face.d < s->nd; face.d++) {
    GfsGradient ng;
    face.neighbor = neighbor (s, cell, face.d);
    face_weighted_gradient (s, &face, &ng, p->u, -1);
    g.a += ng.a;
    g.b += ng.b;
  }
  a *= h*h;
  g.a = 1. + g.a/a;
  g.b = *val (p->rhs, cell) + g.b/a;
  *val (p->res, cell) = g.b - g.a*(*val (p->u, cell));
}

static void diffusion_residual (GtSim * s, Var * u, Var * rhs, Var * rhoc, Var * res)
{
  DiffParams p = { u, rhs, rhoc, res, -1, 0. };
  cell_traverse (s, 0, T_LEAFS, -1, diffusion_residual_cell, &p);
}

/* relax_loop (poisson.c:1070-1089) with diffusion_relax; the homogeneous conditions are those of u */
static void diffusion_relax_loop (GtSim * s, Var * dp, const Var * u, DiffParams * q, unsigned nrelax)
{
  s->bc_homogeneous = 1;
  s->bc_owner = u;
  bc (s, dp, T_LEVEL_LEAFS, q->maxlevel);
  for (unsigned n = 0; n < nrelax - 1; n++) {
    cell_traverse (s, 0, T_LEVEL_LEAFS, q->maxlevel, diffusion_relax, q);
    bc (s, dp, T_LEVEL_LEAFS, q->maxlevel);
  }
  s->bc_homogeneous = 0;
  s->bc_owner = NULL;
  cell_traverse (s, 0, T_LEVEL_LEAFS, q->maxlevel, diffusion_relax, q);
}

/* gfs_diffusion_cycle, poisson.c:1633-1690 */
static void diffusion_cycle (GtSim * s, unsigned levelmin, unsigned depth, unsigned nrelax,
			     Var * u, Var * rhs, Var * rhoc, Var * res)
{
  Var dp;
  var_alloc (s, &dp);
  cell_traverse (s, 1, T_NON_LEAFS, -1, get_from_below_intensive, res);
  DiffParams q = { &dp, NULL, rhoc, res, (int) levelmin, 0. };
  cell_traverse (s, 0, T_LEVEL, levelmin, cell_reset, &dp);
  diffusion_relax_loop (s, &dp, u, &q, 10*nrelax);
  for (q.maxlevel = levelmin + 1; q.maxlevel <= (int) depth; q.maxlevel++) {
    cell_traverse (s, 0, T_LEVEL_NON_LEAFS, q.maxlevel - 1, get_from_above, &dp);
    diffusion_relax_loop (s, &dp, u, &q, nrelax);
  }
  CorrectData cd = { u, &dp };
  cell_traverse (s, 0, T_LEAFS, -1, correct, &cd);
  bc (s, u, T_LEAFS, -1);
  diffusion_residual (s, u, rhs, rhoc, res);
  var_free (s, &dp);
}

/* gfs_domain_norm_variable (domain.c:2197-2232) of the leaves: weights = cell volumes */
typedef struct { Var * v; GoNorm n; } NormVarData;
static void add_norm_variable (GtSim * s, Cell cell, void * data)
{
  NormVarData * p = data;
  double size = cell_size (cell);
  norm_add (&p->n, *val (p->v, cell), s->dim == 3 ? size*size*size : size*size);
}
static GoNorm norm_variable (GtSim * s, Var * v)
{
  NormVarData p = { v, { 0., 0., 0., - DBL_MAX, 0. } };
  cell_traverse (s, 0, T_LEAFS, -1, add_norm_variable, &p);
  norm_update (&p.n);
  return p.n;
}

/* variable_diffusion (timestep.c:923-949) + gfs_diffusion (timestep.c:735-788) */
static void variable_diffusion (GtSim * s, int c, Var * rhs)
{
  GoMultilevelParams * par = &s->diffusion_params[c];
  Var * v = &s->u[c];
  Var rhoc, res;
  var_alloc (s, &rhoc);
  var_alloc (s, &res);
  diffusion_coefficients (s, s->visc[c], s->dt, par->beta, &rhoc);
  DiffParams rp = { v, rhs, &rhoc, NULL, -1, (1. - par->beta)/par->beta };
  cell_traverse (s, 0, T_LEAFS, -1, diffusion_rhs, &rp);
  unsigned minlevel = par->minlevel, maxlevel = s->depth;
  diffusion_residual (s, v, rhs, &rhoc, &res);
  par->residual_before = par->residual = norm_variable (s, &res);
  double res_max_before = par->residual.infty;
  par->niter = 0;
  while (par->niter < par->nitermin ||
	 (par->residual.infty > par->tolerance && par->niter < par->nitermax)) {
    diffusion_cycle (s, minlevel, maxlevel, par->nrelax, v, rhs, &rhoc, &res);
    par->residual = norm_variable (s, &res);
    if (par->residual.infty == res_max_before)
      break;
    if (par->residual.infty > res_max_before/1.1 && minlevel < maxlevel)
      minlevel++;
    res_max_before = par->residual.infty;
    par->niter++;
  }
  var_free (s, &rhoc);
  var_free (s, &res);
}

static void face_reset_un (GtSim * s, const Face * f, void * data) /* advection.c:575-587 */
{
  *val (&s->un[OPP (f->d)], f->neighbor) = *val (&s->un[f->d], f->cell) = 0.;
}

static void face_interpolated_un (GtSim * s, const Face * f, void * data) /* advection.c:549-573 */
{
  double u = face_interpolated_value (s, f, &s->u[f->d/2]);
  *val (&s->un[f->d], f->cell) = u;
  if (!fine_coarse (f))
    *val (&s->un[OPP (f->d)], f->neighbor) = u;
  else
    *val (&s->un[OPP (f->d)], f->neighbor) += u*1./(1.*(s->nc/2) /* FTT_CELLS_DIRECTION */);
}

typedef struct { Var * p, * gv; double dt; } CorrectPar;

static void correct_normal_velocity (GtSim * s, const Face * face, void * data) /* timestep.c:118-144 */
{
  CorrectPar * par = data;
  GfsGradient g;
  face_weighted_gradient (s, face, &g, par->p, -1);
  double dp = (g.b - g.a*(*val (par->p, face->cell)))/cell_size (face->cell);
  if (face->d & 1)
    dp = - dp;
  double f = 1.;
  if (f > 0.)
    dp /= f;
  *val (&s->un[face->d], face->cell) -= dp*par->dt;
  if (par->gv)
    *val (&par->gv[face->d/2], face->cell) += dp*1.;
  if (fine_coarse (face))
    dp *= 1./(1.*s->nc/2);
  *val (&s->un[OPP (face->d)], face->neighbor) -= dp*par->dt;
  if (par->gv)
    *val (&par->gv[face->d/2], face->neighbor) += dp*1.;
}

static void correct_normal_velocities (GtSim * s, Var * p, Var * g, double dt) /* timestep.c:163-179 */
{
  CorrectPar par = { p, g, dt };
  if (s->dim == 2) {
    face_traverse (s, 0, correct_normal_velocity, &par);  /* FTT_XY */
    face_traverse (s, 1, correct_normal_velocity, &par);
  }
  else
    face_traverse (s, -1, correct_normal_velocity, &par); /* FTT_XYZ */
}

static void reset_cell_gradients (GtSim * s, Cell c, void * data) /* timestep.c:36-41 */
{
  Var * g = data;
  for (int k = 0; k < s->dim; k++)
    *val (&g[k], c) = 0.;
}

static void scale_cell_gradients (GtSim * s, Cell cell, void * data) /* timestep.c:60-90 */
{
  Var * g = data;
  for (int c = 0; c < s->dim; c++) {
    Cell c1 = neighbor (s, cell, 2*c), c2 = neighbor (s, cell, 2*c + 1);
    if (exists (c1) && exists (c2))
      *val (&g[c], cell) /= 2.;
  }
}

static void normal_divergence (GtSim * s, Cell cell, void * data) /* fluid.c:2310-2324 */
{
  Var * v = data;
  double div = 0.;
  for (int d = 0; d < s->nd; d++)
    div += ((d & 1) ? -1. : 1.)*(*val (&s->un[d], cell))*1.;
  *val (v, cell) = div*cell_size (cell);
}

typedef struct { Var * div; double dt; } ScalePar;
static void scale_divergence (GtSim * s, Cell cell, void * data) /* timestep.c:181-187 */
{
  ScalePar * p = data;
  *val (p->div, cell) /= p->dt;
}

/* mac_projection, timestep.c:356-444 */
static void mac_projection (GtSim * s, GoMultilevelParams * par, double dt, Var * p, Var * g)
{
  cell_traverse (s, 0, T_LEAFS, -1, reset_cell_gradients, g);
  Var dia, div, res1;
  var_alloc (s, &dia);
  var_alloc (s, &div);
  var_alloc (s, &res1);
  poisson_coefficients (s);
  cell_traverse (s, 0, T_LEAFS, -1, normal_divergence, &div);
  ScalePar sp = { &div, dt };
  cell_traverse (s, 0, T_LEAFS, -1, scale_divergence, &sp);
  poisson_solve (s, par, p, &div, &res1, &dia, dt);
  var_free (s, &dia);
  var_free (s, &div);
  var_free (s, &res1);
  correct_normal_velocities (s, p, g, dt);
  /* gfs_scale_gradients, timestep.c:92-107 */
  cell_traverse (s, 0, T_LEAFS, -1, scale_cell_gradients, g);
  for (int c = 0; c < s->dim; c++)
    bc (s, &g[c], T_LEAFS, -1);
}

typedef struct { Var * g; double dt; } CorrectCentered;
static void correct_centered (GtSim * s, Cell cell, void * data) /* timestep.c:486-496 */
{
  CorrectCentered * p = data;
  for (int c = 0; c < s->dim; c++)
    *val (&s->u[c], cell) -= *val (&p->g[c], cell)*p->dt;
}

static void correct_centered_velocities (GtSim * s, Var * g, double dt) /* timestep.c:498-530 */
{
  CorrectCentered p = { g, dt };
  cell_traverse (s, 0, T_LEAFS, -1, correct_centered, &p);
  for (int c = 0; c < s->dim; c++)
    bc (s, &s->u[c], T_LEAFS, -1);
}

static void approximate_projection (GtSim * s, GoMultilevelParams * par, double dt, Var * p, Var * g)
{ /* timestep.c:560-596 */
  face_traverse (s, -1, face_reset_un, NULL);
  face_traverse (s, -1, face_interpolated_un, NULL);
  mac_projection (s, par, dt, p, g);
  correct_centered_velocities (s, g, dt);
}

/* ---- Godunov advection: advection.c ------------------------------------------------------- */

typedef struct { double dt; Var * v; int use_centered_velocity; int gradient; } AdvPar;

static double transverse_term (GtSim * s, const AdvPar * par, Cell cell, const double * msize, int c)
{ /* advection.c:27-47 */
  double vtan = par->use_centered_velocity ?
    *val (&s->u[c], cell) :
    (*val (&s->un[2*c], cell) + *val (&s->un[2*c + 1], cell))/2.;
  Face f;
  GfsGradient gf;
  f.d = vtan > 0. ? 2*c + 1 : 2*c;
  f.cell = cell;
  f.neighbor = neighbor (s, cell, f.d);
  face_gradient (s, &f, &gf, par->v, -1);
  double g = gf.b - gf.a*(*val (par->v, cell));
  if (vtan > 0.) g = - g;
  return par->dt*vtan*g/(2.*msize[c]);
}

/* source_diffusion_value, source.c:1105-1144 (phi = v, alpha = NULL, constant D): the explicit
   diffusion term of an implicit GfsSourceDiffusion */
static double source_diffusion_value (GtSim * s, const Var * v, Cell cell, double D)
{
  GfsGradient g = { 0., 0. };
  double v0 = *val (v, cell);
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < s->nd; f.d++) {
    f.neighbor = neighbor (s, cell, f.d);
    if (exists (f.neighbor)) {
      GfsGradient e;
      face_gradient (s, &f, &e, v, -1);
      g.a += D*e.a;
      g.b += D*e.b;
    }
  }
  double h = cell_size (cell);
  return 1.*(g.b - g.a*v0)/(h*h);
}

/* gfs_variable_mac_source, source.c:38-59: the sources of v with a mac_value */
static double variable_mac_source (GtSim * s, const Var * v, Cell cell)
{
  for (int c = 0; c < s->dim; c++)
    if (v == &s->u[c] && (s->visc[c] != 0. || s->src[c] != 0.)) {
      double sum = 0.;
      if (s->visc[c] != 0.)
	sum += source_diffusion_value (s, v, cell, s->visc[c]);
      if (s->src[c] != 0.)
	sum += s->src[c];               /* source_value, source.c:398-403 */
      return sum;
    }
  return 0.;
}

static void cell_advected_face_values (GtSim * s, Cell cell, void * data) /* advection.c:58-99 */
{
  const AdvPar * par = data;
  double size = cell_size (cell), msize[3] = { size, size, size };
  for (int c = 0; c < s->dim; c++) {
    double unorm = par->use_centered_velocity ?
      par->dt*(*val (&s->u[c], cell))/msize[c] :
      par->dt*(*val (&s->un[2*c], cell) + *val (&s->un[2*c + 1], cell))/(2.*msize[c]);
    double g = par->gradient ? center_van_leer_gradient (s, cell, c, par->v) :
      center_gradient (s, cell, c, par->v);
    double vl = *val (par->v, cell) + MIN ((1. - unorm)/2., 0.5)*g;
    double vr = *val (par->v, cell) + MAX ((- 1. - unorm)/2., -0.5)*g;
    double src = par->dt*variable_mac_source (s, par->v, cell)/2.;
    double dv;
    if (s->dim == 2)
      dv = transverse_term (s, par, cell, msize, (c + 1) % 2);
    else {
      static const int orthogonal[3][2] = { {1, 2}, {0, 2}, {0, 1} };
      dv =  transverse_term (s, par, cell, msize, orthogonal[c][0]);
      dv += transverse_term (s, par, cell, msize, orthogonal[c][1]);
    }
    *val (&s->fv[2*c], cell)     = vl + src - dv;
    *val (&s->fv[2*c + 1], cell) = vr + src - dv;
  }
}

/* gfs_domain_face_bc (domain.c:1209-1232) on periodic sides (boundary.c:1251-1258,1343-1347): the
   leaf ghost cell beyond side sd holds the face value f[OPP (sd)].v of its periodic image */
typedef struct { int comp, ucomp; const Var * v; } FaceBcPar;
static void face_bc_ghost (GtSim * s, int l, int side, int G, int image, void * data)
{
  if (s->flag[l][G] != GT_LEAF)
    return;
  if (s->side[side] == GO_SIDE_PERIODIC) {
    s->fv[OPP (side)].lev[l][G] = s->fv[OPP (side)].lev[l][image];
    return;
  }
  const FaceBcPar * fp = data;
  int comp = fp->comp, own = ghost_own (s, l, side, G);
  if (fp->ucomp >= 0 && s->bc_u[fp->ucomp][side] == GO_BC_DIRICHLET) {
    /* face_dirichlet, boundary.c:270-275 */
    s->fv[OPP (side)].lev[l][G] = s->fv[side].lev[l][own] = s->bcu[fp->ucomp].lev[l][G];
    return;
  }
  if (fp->ucomp >= 0 && s->bc_u[fp->ucomp][side] == GO_BC_NEUMANN) {
    /* face_neumann, boundary.c:349-355 */
    s->fv[OPP (side)].lev[l][G] = fp->v->lev[l][own] + s->bcu[fp->ucomp].lev[l][G]*(1./s->n[l])/2.;
    return;
  }
  /* face_symmetry, boundary.c:64-74 */
  if (comp == side/2)
    s->fv[OPP (side)].lev[l][G] = s->fv[side].lev[l][own] = 0.;
  else
    s->fv[OPP (side)].lev[l][G] = s->fv[side].lev[l][own];
}

static void face_bc (GtSim * s, const Var * v)
{
  FaceBcPar fp = { var_component (s, v), -1, v };
  for (int c = 0; c < 3; c++)
    if (v == &s->u[c]) fp.ucomp = c;
  for (int l = 0; l <= s->depth; l++)
    ghost_traverse (s, l, face_bc_ghost, &fp);
}

static void face_values_set (GtSim * s, AdvPar * par) /* timestep.c:644-654 */
{
  cell_traverse (s, 0, T_LEAFS, -1, cell_advected_face_values, par);
  face_bc (s, par->v);
}

static int is_interior (const GtSim * s, Cell c)   /* !GFS_CELL_IS_BOUNDARY */
{
  int i = cell_i (s, c), j = cell_j (s, c), k = cell_k (s, c), n = s->n[c.l];
  return i >= 1 && i <= n && j >= 1 && j <= n && k >= 1 && k <= n;
}

/* interpolate_1D1 of advection.c:132-180.  The fork's text declares s2 twice (the second, inner
   declaration shadows the variable the two assignments above it set); the restatement takes the
   assigned values, which is upstream Gerris' behaviour -- the one that produced the .ref files. */
static double adv_interpolate_1D1 (const GtSim * s, Cell cell, int dright, int dup, double x)
{
  int dleft = OPP (dright);
  Cell n = neighbor (s, cell, dup);
  if (exists (n) && is_interior (s, n)) {
    double s2 = is_leaf (s, n) ? 1. : 0.5;
    double s1 = 1.;
    double v1 = *val (&s->fv[dleft], cell), v2;
    assert (n.l == cell.l);
    if (is_leaf (s, n))
      v2 = *val (&s->fv[dleft], n);
    else {
      n = child_corner (s, n, dleft, OPP (dup), -1);
      if (exists (n))
	v2 = *val (&s->fv[dleft], n);
      else
	s2 = v2 = 0.;
    }
    return s2 > 0. ? (v2*(s1 - 1. + 2.*x) + v1*(s2 + 1. - 2.*x))/(s1 + s2) : v1;
  }
  return *val (&s->fv[dleft], cell);
}

/* interpolate_2D1 of advection.c:183-249 (3-D) */
static double adv_interpolate_2D1 (const GtSim * s, Cell cell, int dright, int d1, int d2,
				   double x, double y)
{
  double x1 = 0., y1 = 1.;
  double x2 = 1., y2 = 0.;
  double v0, v1, v2;
  int dleft = OPP (dright);
  v0 = *val (&s->fv[dleft], cell);
  Cell n1 = neighbor (s, cell, d1);
  if (exists (n1) && is_interior (s, n1)) {
    assert (n1.l == cell.l);
    if (!is_leaf (s, n1)) {
      n1 = child_corner (s, n1, OPP (dright), OPP (d1), d2);
      if (exists (n1)) {
	v1 = *val (&s->fv[dleft], n1);
	x1 = 1./4.;
	y1 = 3./4.;
      }
      else
	v1 = v0;
    }
    else
      v1 = *val (&s->fv[dleft], n1);
  }
  else
    v1 = v0;
  Cell n2 = neighbor (s, cell, d2);
  if (exists (n2) && is_interior (s, n2)) {
    assert (n2.l == cell.l);
    if (!is_leaf (s, n2)) {
      n2 = child_corner (s, n2, OPP (dright), OPP (d2), d1);
      if (exists (n2)) {
	v2 = *val (&s->fv[dleft], n2);
	x2 = 3./4.;
	y2 = 1./4.;
      }
      else
	v2 = v0;
    }
    else
      v2 = *val (&s->fv[dleft], n2);
  }
  else
    v2 = v0;
  return ((v1 - v0)*(x*y2 - x2*y) + (v2 - v0)*(x1*y - x*y1))/(x1*y2 - x2*y1) + v0;
}

/* gfs_face_upwinded_value, advection.c:267-343 */
static double face_upwinded_value (GtSim * s, const Face * face, int centered_upwinding)
{
  double un;
  if (centered_upwinding)
    un = face_interpolated_value (s, face, &s->u[face->d/2]);
  else
    un = *val (&s->un[face->d], face->cell);
  if (face->d & 1)
    un = - un;
  double fc = *val (&s->fv[face->d], face->cell);
  if (!fine_coarse (face)) {
    double fn = *val (&s->fv[OPP (face->d)], face->neighbor);
    return un > 0. ? fc : un < 0. ? fn : (fc + fn)/2.;
  }
  if (un > 0.)
    return fc;
  double vcoarse;
  if (s->dim == 2) {
    int dp = perpendicular2[face->d][cell_id (s, face->cell)];
    assert (dp >= 0);
    vcoarse = adv_interpolate_1D1 (s, face->neighbor, face->d, dp, 1./4.);
  }
  else {
    const int * dp = perpendicular3[face->d][cell_id (s, face->cell)];
    assert (dp[0] >= 0 && dp[1] >= 0);
    vcoarse = adv_interpolate_2D1 (s, face->neighbor, face->d, dp[0], dp[1], 1./4., 1./4.);
  }
  if (un == 0.)
    return (fc + vcoarse)/2.;
  return vcoarse;
}

static void face_advected_normal_velocity (GtSim * s, const Face * face, void * data)
{ /* advection.c:513-539 */
  double u = face_upwinded_value (s, face, 1);
  *val (&s->un[face->d], face->cell) = u;
  if (!fine_coarse (face))
    *val (&s->un[OPP (face->d)], face->neighbor) = u;
  else
    *val (&s->un[OPP (face->d)], face->neighbor) += u*1./(1.*(s->nc/2));
}

static void predicted_face_velocities (GtSim * s) /* timestep.c:681-717 */
{
  face_traverse (s, -1, face_reset_un, NULL);
  AdvPar par = { s->dt, NULL, 1, 0 };
  for (int c = 0; c < s->dim; c++) {
    par.v = &s->u[c];
    face_values_set (s, &par);
    face_traverse (s, c, face_advected_normal_velocity, NULL);
  }
}

typedef struct { double dt; Var * fvar, * g; int c; } FluxPar;

static void face_reset (GtSim * s, const Face * f, void * data) /* fluid.c gfs_face_reset */
{
  FluxPar * p = data;
  *val (p->fvar, f->cell) = *val (p->fvar, f->neighbor) = 0.;
}

static void face_velocity_advection_flux (GtSim * s, const Face * face, void * data)
{ /* advection.c:398-435 */
  FluxPar * par = data;
  double flux = 1.*(*val (&s->un[face->d], face->cell))*par->dt/cell_size (face->cell);
  flux *= face_upwinded_value (s, face, 0)
    - face_interpolated_value (s, face, &par->g[par->c])*par->dt/2.;
  if (face->d & 1)
    flux = - flux;
  *val (par->fvar, face->cell) -= flux;
  if (!fine_coarse (face))
    *val (par->fvar, face->neighbor) += flux;
  else
    *val (par->fvar, face->neighbor) += flux/s->nc /* FTT_CELLS */;
}

static void face_advection_flux (GtSim * s, const Face * face, void * data)
{ /* gfs_face_advection_flux, advection.c:356-381 */
  FluxPar * par = data;
  double flux = 1.*(*val (&s->un[face->d], face->cell))*par->dt*
    face_upwinded_value (s, face, 0)/cell_size (face->cell);
  if (face->d & 1)
    flux = - flux;
  *val (par->fvar, face->cell) -= flux;
  if (!fine_coarse (face))
    *val (par->fvar, face->neighbor) += flux;
  else
    *val (par->fvar, face->neighbor) += flux/s->nc /* FTT_CELLS */;
}

typedef struct { Var * sv, * fvar, * g; double dt; } UpdatePar;
static void advection_update (GtSim * s, Cell cell, void * data) /* advection.c:784-819 */
{
  UpdatePar * p = data;
  *val (p->sv, cell) += *val (p->fvar, cell)/1.;
}
static void add_pressure_gradient (GtSim * s, Cell cell, void * data) /* timestep.c:809-812 */
{
  UpdatePar * p = data;
  *val (p->sv, cell) -= *val (p->g, cell)*p->dt;
}

typedef struct { Var * sv; double dt, g; } SrcData;
static void add_centered_source (GtSim * s, Cell c, void * data)
{
  SrcData * d = data;
  double sum = 0;
  sum += d->g;
  *val (d->sv, c) += d->dt*sum;
}

/* variable_sources, timestep.c:872-921, for a velocity component */
static void variable_sources (GtSim * s, int c, Var * sv, double dt, Var * gmac, Var * g)
{
  Var fvar;
  var_alloc (s, &fvar);
  FluxPar fp = { dt, &fvar, gmac, c };
  AdvPar ap = { dt, &s->u[c], 0, 0 };
  face_traverse (s, -1, face_reset, &fp);
  face_values_set (s, &ap);
  face_traverse (s, -1, face_velocity_advection_flux, &fp);
  UpdatePar up = { sv, &fvar, g ? &g[c] : NULL, dt };
  cell_traverse (s, 0, T_LEAFS, -1, advection_update, &up);
  var_free (s, &fvar);
  if (g)
    cell_traverse (s, 0, T_LEAFS, -1, add_pressure_gradient, &up);
  if (s->src[c] != 0.) {
    /* gfs_domain_variable_centered_sources, source.c:62-108: the sources with a centered_value (a GfsSource) */
    SrcData sd = { sv, dt, s->src[c] };
    cell_traverse (s, 0, T_LEAFS, -1, add_centered_source, &sd);
  }
}

/* gfs_tracer_advection_diffusion (timestep.c:1028-1055, no diffusion) with variable_sources :872-921:
   gfs_face_advection_flux, the gradient of the GfsVariableTracer, then gfs_domain_bc */
static void tracer_advection (GtSim * s, int t, double dt)
{
  Var fvar;
  var_alloc (s, &fvar);
  FluxPar fp = { dt, &fvar, NULL, -1 };
  AdvPar ap = { dt, &s->tracer[t], 0, s->tracer_gradient[t] };
  face_traverse (s, -1, face_reset, &fp);
  face_values_set (s, &ap);
  face_traverse (s, -1, face_advection_flux, &fp);
  UpdatePar up = { &s->tracer[t], &fvar, NULL, dt };
  cell_traverse (s, 0, T_LEAFS, -1, advection_update, &up);
  var_free (s, &fvar);
  bc (s, &s->tracer[t], T_LEAFS, -1);
}

static void advance_tracers (GtSim * s, double dt) /* simulation.c:405-430 */
{
  for (int t = 0; t < s->ntracers; t++)
    tracer_advection (s, t, dt);
}

typedef struct { Var * dst; const Var * src; } CopyData;
static void copy_v_rhs (GtSim * s, Cell c, void * data) /* timestep.c:951-954 */
{
  CopyData * d = data;
  *val (d->dst, c) = *val (d->src, c);
}

static void centered_velocity_advection (GtSim * s, Var * gmac, Var * g) /* timestep.c:976-1016 */
{
  for (int c = 0; c < s->dim; c++)
    if (s->visc[c] != 0.) {
      /* source_diffusion (v[c]): rhs = copy of v, the sources into rhs, then the implicit solve */
      Var rhs;
      var_alloc (s, &rhs);
      CopyData cd = { &rhs, &s->u[c] };
      cell_traverse (s, 0, T_LEAFS, -1, copy_v_rhs, &cd);
      variable_sources (s, c, &rhs, s->dt, gmac, g);
      variable_diffusion (s, c, &rhs);
      var_free (s, &rhs);
    }
    else
      variable_sources (s, c, &s->u[c], s->dt, gmac, g);
  for (int c = 0; c < s->dim; c++)
    bc (s, &s->u[c], T_LEAFS, -1);
}

/* ---- CFL, time step, coarse values ----------------------------------------------------------- */

static void minimum_mac_cfl (GtSim * s, const Face * face, void * data) /* domain.c:2824-2856 */
{
  double * cfl = data;
  double un = *val (&s->un[face->d], face->cell);
  double length = cell_size (face->cell);
  if (un != 0.) {
    double cflu = length/fabs (un);
    if (cflu*cflu < *cfl)
      *cfl = cflu*cflu;
  }
}

static void minimum_cfl (GtSim * s, Cell cell, void * data) /* domain.c:2858-2897 */
{
  double * cfl = data;
  double length = cell_size (cell);
  for (int c = 0; c < s->dim; c++) {
    double fm = 1.;
    if (*val (&s->u[c], cell) != 0.) {
      double cflu = length/fabs (fm*(*val (&s->u[c], cell)));
      if (cflu*cflu < *cfl)
	*cfl = cflu*cflu;
    }
    if (s->visc[c] != 0. || s->src[c] != 0.) {       /* p->v[c]->sources, domain.c:2882-2891 */
      double g = variable_mac_source (s, &s->u[c], cell);
      if (g != 0.) {
	double cflg = 2.*length/fabs (fm*g);
	if (cflg < *cfl)
	  *cfl = cflg;
      }
    }
  }
}

static double domain_cfl (GtSim * s) /* domain.c:2899-2923 */
{
  double cfl = DBL_MAX;
  face_traverse (s, -1, minimum_mac_cfl, &cfl);
  cell_traverse (s, 0, T_LEAFS, -1, minimum_cfl, &cfl);
  return sqrt (cfl);
}

static void set_timestep (GtSim * s) /* simulation.c:1569-1633; the only event time is `end' */
{
  double t = s->t;
  s->dt = s->cfl*domain_cfl (s);
  double tnext = G_MAXINT;
  if (s->end < tnext)
    tnext = s->end;
  double n = ceil ((tnext - t)/s->dt);
  if (n > 0. && n < G_MAXINT) {
    s->dt = (tnext - t)/n;
    if (n == 1.)
      s->tnext = tnext;
    else
      s->tnext = t + s->dt;
  }
  else
    s->tnext = t + s->dt;
  if (s->dt < 1e-9)
    s->dt = 1e-9;
}

static void get_from_below_intensive (GtSim * s, Cell cell, void * data) /* fluid.c:1843-1864 */
{
  Var * v = data;
  double sum = 0., sa = 0.;
  for (int k = 0; k < s->nc; k++) {
    Cell ch = child (s, cell, k);
    if (exists (ch)) {
      double a = 1.;
      sum += *val (v, ch)*a;
      sa += a;
    }
  }
  *val (v, cell) = sum/sa;
}

static void coarse_init (GtSim * s) /* adaptive.c:43-58 on every variable */
{
  Var * all[] = { &s->p, &s->pmac, &s->u[0], &s->u[1], &s->u[2] };
  for (int k = 0; k < 2 + s->dim; k++)
    cell_traverse (s, 1, T_NON_LEAFS, -1, get_from_below_intensive, all[k]);
  for (int t = 0; t < s->ntracers; t++)
    cell_traverse (s, 1, T_NON_LEAFS, -1, get_from_below_intensive, &s->tracer[t]);
}

/* ---- the simulation of test/periodic/periodic.gfs ------------------------------------------ */

static void cell_pos (const GtSim * s, Cell c, double * x, double * y) /* ftt_cell_pos, box = unit square */
{
  double h = cell_size (c);
  *x = -0.5 + (cell_i (s, c) - 0.5)*h;
  *y = -0.5 + (cell_j (s, c) - 0.5)*h;
}

static void init_uv (GtSim * s, Cell c, void * data) /* periodic.gfs:26-29 */
{
  double x, y;
  cell_pos (s, c, &x, &y);
  *val (&s->u[0], c) = (1. - 2.*cos (2.*M_PI*x)*sin (2.*M_PI*y));
  *val (&s->u[1], c) = (1. + 2.*sin (2.*M_PI*x)*cos (2.*M_PI*y));
}

/* a GfsSimulation on one periodic box refined by `refine' (default parameters of
   gfs_multilevel_params_init / gfs_advection_params_init; U, V, P zero) */
static GtSim * gt_new_sides (int dim, GtRefineFunc refine, void * ctx, const int * side);

GtSim * gt_new (int dim, GtRefineFunc refine, void * ctx)
{
  return gt_new_sides (dim, refine, ctx, NULL);
}

/* side[d] = GO_SIDE_PERIODIC or GO_SIDE_BOUNDARY (NULL: all periodic) */
GtSim * gt_new_with_sides (int dim, GtRefineFunc refine, void * ctx, const int * side)
{
  return gt_new_sides (dim, refine, ctx, side);
}

static GtSim * gt_new_sides (int dim, GtRefineFunc refine, void * ctx, const int * side)
{
  GtSim * s = calloc (1, sizeof (GtSim));
  assert (dim == 2 || dim == 3);
  s->dim = dim;
  s->nd = 2*dim;
  s->nc = 1 << dim;
  for (int d = 0; d < 6; d++) {
    s->side[d] = side && d < 2*dim ? side[d] : GO_SIDE_PERIODIC;
    s->bc_p[d] = GO_BC_SYMMETRY;
  }
  build_tree (s, refine, ctx);
  var_alloc (s, &s->bcval);
  Var * all[] = { &s->p, &s->pmac, &s->u[0], &s->u[1], &s->u[2], &s->g[0], &s->g[1], &s->g[2],
		  &s->gmac[0], &s->gmac[1], &s->gmac[2] };
  for (unsigned k = 0; k < sizeof (all)/sizeof (all[0]); k++)
    var_alloc (s, all[k]);
  for (int d = 0; d < 6; d++) {
    var_alloc (s, &s->un[d]);
    var_alloc (s, &s->fv[d]);
    var_alloc (s, &s->w[d]);
  }
  go_multilevel_params_init (&s->projection_params, dim);
  go_multilevel_params_init (&s->approx_projection_params, dim);
  for (int c = 0; c < 3; c++) {     /* diffusion_init, source.c:966-974 */
    go_multilevel_params_init (&s->diffusion_params[c], dim);
    s->diffusion_params[c].tolerance = 1e-6;
  }
  s->cfl = 0.8;
  s->end = DBL_MAX;
  return s;
}

void gt_set_time (GtSim * s, double end, double cfl)
{
  s->end = end;
  s->cfl = cfl;
}

/* the Refine function of test/periodic/periodic.gfs:25 */
typedef struct { int level, box; } PeriodicRefine;
static double periodic_refine (double x, double y, double z, void * ctx)
{
  PeriodicRefine * p = ctx;
  return (x < -0.25 || x > 0.25 || y < -0.25 || y > 0.25 ? p->level : p->level + p->box);
}

/* test/periodic/periodic.gfs with LEVEL = level, BOX = box */
GtSim * gt_periodic_new (int level, int box)
{
  PeriodicRefine pr = { level, box };
  GtSim * s = gt_new (2, periodic_refine, &pr);
  s->projection_params.tolerance = 1e-6;        /* periodic.gfs:30-31 */
  s->approx_projection_params.tolerance = 1e-6;
  s->cfl = 0.75;                                /* periodic.gfs:24 */
  s->end = 0.5;                                 /* periodic.gfs:23 */
  cell_traverse (s, 0, T_LEAFS, -1, init_uv, NULL);
  return s;
}

void gt_destroy (GtSim * s)
{
  Var * all[] = { &s->p, &s->pmac, &s->u[0], &s->u[1], &s->u[2], &s->g[0], &s->g[1], &s->g[2],
		  &s->gmac[0], &s->gmac[1], &s->gmac[2] };
  for (unsigned k = 0; k < sizeof (all)/sizeof (all[0]); k++)
    var_free (s, all[k]);
  for (int d = 0; d < 6; d++) {
    var_free (s, &s->un[d]);
    var_free (s, &s->fv[d]);
    var_free (s, &s->w[d]);
  }
  var_free (s, &s->bcval);
  for (int t = 0; t < s->ntracers; t++)
    var_free (s, &s->tracer[t]);
  if (s->bcu_alloc)
    for (int q = 0; q < 3; q++)
      var_free (s, &s->bcu[q]);
  for (int l = 0; l <= s->depth; l++)
    free (s->flag[l]);
  free (s);
}

/* simulation_run up to the loop, simulation.c:458-476 */
void gt_start (GtSim * s)
{
  Var * all[] = { &s->p, &s->pmac, &s->u[0], &s->u[1], &s->u[2] };
  for (int k = 0; k < 2 + s->dim; k++)
    bc (s, all[k], T_LEAFS, -1);
  for (int t = 0; t < s->ntracers; t++)
    bc (s, &s->tracer[t], T_LEAFS, -1);
  coarse_init (s);
  set_timestep (s);
  approximate_projection (s, &s->approx_projection_params, s->dt, &s->p, s->g);
  set_timestep (s);
  advance_tracers (s, s->dt/2.);
}

/* one iteration of the loop, simulation.c:479-548 */
void gt_step (GtSim * s)
{
  predicted_face_velocities (s);
  mac_projection (s, &s->projection_params, s->dt/2., &s->pmac, s->gmac); /* p <-> pmac swapped */
  centered_velocity_advection (s, s->gmac, s->i > 0 ? s->g : s->gmac);
  correct_centered_velocities (s, s->i > 0 ? s->g : s->gmac, - s->dt);
  coarse_init (s);
  approximate_projection (s, &s->approx_projection_params, s->dt, &s->p, s->g);
  s->t = s->tnext;
  s->i++;
  set_timestep (s);
  advance_tracers (s, s->dt);
}

/* conditions of velocity component c on side d: GO_BC_SYMMETRY (default) / _DIRICHLET / _NEUMANN; the
   values per ghost cell (the GfsFunction at the face centre) in gt_bc_values_u (s, c, l) */
void gt_set_bc_u (GtSim * s, int c, int d, int kind)
{
  if (!s->bcu_alloc) {
    for (int q = 0; q < 3; q++)
      var_alloc (s, &s->bcu[q]);
    s->bcu_alloc = 1;
  }
  s->bc_u[c][d] = kind;
}
double * gt_bc_values_u (GtSim * s, int c, int l)
{
  if (!s->bcu_alloc) {
    for (int q = 0; q < 3; q++)
      var_alloc (s, &s->bcu[q]);
    s->bcu_alloc = 1;
  }
  return s->bcu[c].lev[l];
}
/* GfsSourceDiffusion {} U|V|W nu, and its GfsMultilevelParams (tolerance 1e-6: diffusion_init,
   source.c:966-974) */
void gt_set_viscosity (GtSim * s, int c, double nu) { s->visc[c] = nu; }
void gt_set_source (GtSim * s, int c, double g) { s->src[c] = g; }
GoMultilevelParams * gt_diffusion_params (GtSim * s, int c) { return &s->diffusion_params[c]; }

/* GfsVariableTracer T [{ gradient = ... }]: returns the index of gt_values (17 + t) */
int gt_add_tracer (GtSim * s, int gradient)
{
  if (s->ntracers >= GT_MAXTRACERS)
    return -1;
  var_alloc (s, &s->tracer[s->ntracers]);
  s->tracer_gradient[s->ntracers] = gradient;
  return 17 + s->ntracers++;
}

double gt_time (const GtSim * s) { return s->t; }
double gt_end (const GtSim * s) { return s->end; }
unsigned gt_iter (const GtSim * s) { return s->i; }
int gt_depth (const GtSim * s) { return s->depth; }
double gt_dt (const GtSim * s) { return s->dt; }
GoMultilevelParams * gt_projection_params (GtSim * s, int approx)
{ return approx ? &s->approx_projection_params : &s->projection_params; }

int gt_dim (const GtSim * s) { return s->dim; }

/* number of leaves per level (tree checks) */
void gt_leaf_count (const GtSim * s, long * count)
{
  for (int l = 0; l <= s->depth; l++) {
    count[l] = 0;
    for (int k = 1; k <= (s->dim == 3 ? s->n[l] : 1); k++)
      for (int j = 1; j <= s->n[l]; j++)
	for (int i = 1; i <= s->n[l]; i++)
	  if (s->flag[l][i + s->r[l]*(j + (s->dim == 3 ? s->r[l]*k : 0))] == GT_LEAF)
	    count[l]++;
  }
}

/* flags / values of a level for the tests: which = 0 U, 1 V, 2 P, 3 Pmac, 4-5 g, 6-7 gmac, 8-11 un[0..3],
   12 W, 13 g[2], 14 gmac[2], 15-16 un[4..5] */
const unsigned char * gt_flags (const GtSim * s, int l) { return s->flag[l]; }
double * gt_values (GtSim * s, int which, int l)
{
  Var * v[] = { &s->u[0], &s->u[1], &s->p, &s->pmac, &s->g[0], &s->g[1], &s->gmac[0], &s->gmac[1],
		&s->un[0], &s->un[1], &s->un[2], &s->un[3], &s->u[2], &s->g[2], &s->gmac[2],
		&s->un[4], &s->un[5], &s->tracer[0], &s->tracer[1] };
  return v[which]->lev[l];
}

/* GfsOutputErrorNorm { v = U } { s = 1 - 2 cos (2 pi (x - t)) sin (2 pi (y - t)) } (periodic.gfs:32-34,
   output.c:2953-3013, domain.c:2116-2122): norms of U - s over the leaves, weight = cell volume */
typedef struct { GoNorm n; double t; } ErrPar;
static void add_error (GtSim * s, Cell c, void * data)
{
  ErrPar * p = data;
  double x, y, h = cell_size (c);
  cell_pos (s, c, &x, &y);
  double ref = (1. - 2.*cos (2.*M_PI*(x - p->t))*sin (2.*M_PI*(y - p->t)));
  norm_add (&p->n, *val (&s->u[0], c) - ref, s->dim == 3 ? h*h*h : h*h);   /* gfs_cell_volume */
}

void gt_error_norm (GtSim * s, double * first, double * second, double * infty)
{
  ErrPar p = { { 0., 0., 0., - DBL_MAX, 0. }, s->t };
  cell_traverse (s, 0, T_LEAFS, -1, add_error, &p);
  norm_update (&p.n);
  *first = p.n.first;
  *second = p.n.second;
  *infty = p.n.infty;
}

/* ---- GfsPoisson: poisson_run, simulation.c:2150-2285 ---------------------------------------- */

void gt_set_bc (GtSim * s, int d, int kind) { s->bc_p[d] = kind; }
double * gt_bc_values (GtSim * s, int l) { return s->bcval.lev[l]; }

typedef struct { Var * divu, * div; double sum_div, sum_vol, ddiv; long n; } DivData;

static void rescale_div (GtSim * s, Cell c, void * data) /* simulation.c:2156-2162 */
{
  DivData * p = data;
  double size = cell_size (c);
  double a = size*size*1.;
  *val (p->div, c) = *val (p->divu, c)*a;
  p->sum_vol += a;        /* gts_range_add_value: mean = sum/n */
  p->n++;
}

static void sum_div (GtSim * s, Cell c, void * data) { ((DivData *) data)->sum_div += *val (((DivData *) data)->div, c); }

static void add_ddiv (GtSim * s, Cell c, void * data) /* simulation.c:2164-2168 */
{
  DivData * p = data;
  double size = cell_size (c);
  *val (p->div, c) += size*size*p->ddiv*1.;
}

/* one iteration of poisson_run: `divu' holds the variable Div on the leaves, P the current guess;
   the solve uses the approximate-projection parameters.  The residual is left in `res'. */
void gt_poisson_run (GtSim * s, Var * divu_unused)
{
  (void) divu_unused;
  Var div, dia, res;
  var_alloc (s, &div); var_alloc (s, &dia); var_alloc (s, &res);
  int dirichlet = 0;
  for (int d = 0; d < s->nd; d++)
    if (s->side[d] == GO_SIDE_BOUNDARY && s->bc_p[d] == GO_BC_DIRICHLET)
      dirichlet = 1;
  /* gfs_simulation_init: the BCs of every variable; gfs_cell_coarse_init */
  s->bc_homogeneous = 0;
  bc (s, &s->p, T_LEAFS, -1);
  cell_traverse (s, 1, T_NON_LEAFS, -1, get_from_below_intensive, &s->p);
  /* correct_div, simulation.c:2170-2190: the variable Div is kept in pmac here */
  DivData dd = { &s->pmac, &div, 0., 0., 0., 0 };
  cell_traverse (s, 0, T_LEAFS, -1, rescale_div, &dd);
  if (!dirichlet) {
    cell_traverse (s, 0, T_LEAFS, -1, sum_div, &dd);
    dd.ddiv = - (dd.sum_div/dd.n)/(dd.sum_vol/dd.n);
    cell_traverse (s, 0, T_LEAFS, -1, add_ddiv, &dd);
  }
  for (int l = 0; l <= s->depth; l++)     /* kept for the tests: the right-hand side of the solve (slot g[0]) */
    memcpy (s->g[0].lev[l], div.lev[l], s->size[l]*sizeof (double));
  poisson_coefficients (s);
  poisson_solve (s, &s->approx_projection_params, &s->p, &div, &res, &dia, 1.);
  s->i++;
  var_free (s, &div); var_free (s, &dia); var_free (s, &res);
}

/* gfs_face_interpolated_value_generic, fluid.c:2200-2221 */
static double face_interpolated_value_generic (const GtSim * s, const Face * face, const Var * v)
{
  if (!exists (face->neighbor) || is_leaf (s, face->neighbor) || face->neighbor.l < face->cell.l)
    return face_interpolated_value (s, face, v);
  /* finer neighbor */
  Face f = { NOCELL, face->cell, OPP (face->d) };
  Cell ch[4];
  int n = children_direction (s, face->neighbor, f.d, ch);
  double avg = 0.;
  for (int i = 0; i < n; i++)
    if (exists (ch[i])) {
      f.cell = ch[i];
      avg += face_interpolated_value (s, &f, v)*1.;
    }
  return avg == 0. ? 0. : avg/(1.*n);
}

/* gfs_divergence, fluid.c:2357-2376: the derived variable `Divergence' of a leaf */
static double divergence (const GtSim * s, Cell cell)
{
  double div = 0.;
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < s->nd; f.d++) {
    f.neighbor = neighbor (s, cell, f.d);
    if (exists (f.neighbor))
      div += 1.*((f.d & 1) ? -1. : 1.)*face_interpolated_value_generic (s, &f, &s->u[f.d/2]);
  }
  return div/(1.*cell_size (cell));
}

typedef struct { GoNorm n; double sum; } DivPar;
static void add_divergence (GtSim * s, Cell c, void * data)
{
  DivPar * p = data;
  double h = cell_size (c), vol = s->dim == 3 ? h*h*h : h*h;
  norm_add (&p->n, divergence (s, c), vol);
  double v2 = 0.;
  for (int k = 0; k < s->dim; k++)
    v2 += *val (&s->u[k], c)*(*val (&s->u[k], c));
  p->sum += vol*v2;
}

/* GfsOutputScalarNorm { v = Divergence } (output.c:1966-1986, domain.c:2197-2232: weights = cell
   volumes) and GfsOutputScalarSum { v = Velocity2 } (output.c:2089-2123) over the leaves */
void gt_divergence_norm (GtSim * s, double * first, double * second, double * infty, double * velocity2_sum)
{
  DivPar p = { { 0., 0., 0., - DBL_MAX, 0. }, 0. };
  cell_traverse (s, 0, T_LEAFS, -1, add_divergence, &p);
  norm_update (&p.n);
  *first = p.n.first;
  *second = p.n.second;
  *infty = p.n.infty;
  *velocity2_sum = p.sum;
}

/* the value of `Divergence' on the leaves of a level (other cells: 0) */
void gt_divergence_level (GtSim * s, int l, double * out)
{
  for (size_t q = 0; q < s->size[l]; q++) {
    Cell c = { l, (int) q };
    out[q] = s->flag[l][q] == GT_LEAF && is_interior (s, c) ? divergence (s, c) : 0.;
  }
}

/* the whole run: returns the number of steps */
unsigned gt_run (GtSim * s)
{
  gt_start (s);
  while (s->t < s->end)
    gt_step (s);
  return s->i;
}
