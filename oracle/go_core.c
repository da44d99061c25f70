/* go_core.c -- oracle: domain, traversal order, boundary conditions.
 * TEST INFRASTRUCTURE ONLY (see gfs_oracle.h). */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "gfs_oracle.h"

/* Child n of a cell sits at x:+ for bit0, y:- for bit1, z:- for bit2
 * (coords[] table, ftt.c:301-316); children are visited n = 0..FTT_CELLS-1
 * (cell_traverse_level_leafs, ftt.c:837-852).  (pi,pj,pk) are 0-based cell coordinates
 * at `level` with j growing with y and k with z. */
static void order_rec (const GoDomain * dom, int level, int target,
		       int pi, int pj, int pk, int * out, int * cnt)
{
  if (level == target) {
    out[(*cnt)++] = (int) go_index (dom, target, pi + 1, pj + 1, dom->dim == 3 ? pk + 1 : 0);
    return;
  }
  int ncells = 1 << dom->dim;
  for (int n = 0; n < ncells; n++) {
    int ci = 2*pi + (n & 1);
    int cj = 2*pj + ((n & 2) ? 0 : 1);
    int ck = dom->dim == 3 ? 2*pk + ((n & 4) ? 0 : 1) : 0;
    order_rec (dom, level + 1, target, ci, cj, ck, out, cnt);
  }
}

size_t go_level_size (const GoDomain * dom, int level)
{
  return dom->size[level];
}

size_t go_index (const GoDomain * dom, int level, int i, int j, int k)
{
  size_t s = dom->n[level] + 2;
  return dom->dim == 3 ? i + s*(j + s*k) : i + s*j;
}

const int * go_order (const GoDomain * dom, int level)
{
  return dom->order[level];
}

/* is the cell at linear index `idx` adjacent to side d? */
static int on_side (const GoDomain * dom, int level, int idx, int d)
{
  int s = dom->n[level] + 2, n = dom->n[level];
  int c[3] = { idx % s, (idx/s) % s, dom->dim == 3 ? idx/(s*s) : 1 };
  return (d & 1) ? c[d/2] == 1 : c[d/2] == n;
}

GoDomain * go_domain_new (int dim, int depth, const int side[6])
{
  assert (dim == 2 || dim == 3);
  assert (depth >= 0 && depth <= GO_MAXLEVEL);
  GoDomain * dom = calloc (1, sizeof (GoDomain));
  dom->dim = dim;
  dom->depth = depth;
  for (int d = 0; d < 6; d++)
    dom->side[d] = side ? side[d] : GO_SIDE_BOUNDARY;
  for (int l = 0; l <= depth; l++) {
    int n = 1 << l;
    size_t s = n + 2;
    dom->n[l] = n;
    dom->size[l] = dim == 3 ? s*s*s : s*s;
    dom->off[l][0] = 1;  dom->off[l][1] = -1;
    dom->off[l][2] = s;  dom->off[l][3] = -(ptrdiff_t) s;
    dom->off[l][4] = s*s; dom->off[l][5] = -(ptrdiff_t) (s*s);
    size_t ncell = dim == 3 ? (size_t) n*n*n : (size_t) n*n;
    dom->order[l] = malloc (ncell*sizeof (int));
    int cnt = 0;
    order_rec (dom, 0, l, 0, 0, 0, dom->order[l], &cnt);
    assert ((size_t) cnt == ncell);
    /* cells adjacent to each side, in traversal order (ftt_cell_traverse_boundary, ftt.c) */
    size_t nb = dim == 3 ? (size_t) n*n : (size_t) n;
    dom->nborder[l] = (int) nb;
    for (int d = 0; d < 2*dim; d++) {
      dom->border[l][d] = malloc (nb*sizeof (int));
      size_t m = 0;
      for (size_t q = 0; q < ncell; q++)
	if (on_side (dom, l, dom->order[l][q], d))
	  dom->border[l][d][m++] = dom->order[l][q];
      assert (m == nb);
    }
    for (int d = 0; d < 2*dim; d++)
      dom->w[d][l] = calloc (dom->size[l], sizeof (double));
  }
  return dom;
}

void go_domain_destroy (GoDomain * dom)
{
  if (!dom) return;
  for (int l = 0; l <= dom->depth; l++) {
    free (dom->order[l]);
    for (int d = 0; d < 6; d++) {
      free (dom->border[l][d]);
      free (dom->w[d][l]);
    }
  }
  free (dom);
}

void go_domain_set_hooks (GoDomain * dom, GoExchangeFunc ex, void * ex_ctx,
			  GoReduceFunc red, void * red_ctx)
{
  dom->exchange = ex; dom->exchange_ctx = ex_ctx;
  dom->reduce = red; dom->reduce_ctx = red_ctx;
}

/* GFS_STATE (cell)->f[d].v of the cells of a level, as set by go_poisson_coefficients* */
double * go_domain_weight (GoDomain * dom, int d, int level)
{
  return dom->w[d][level];
}

/* the domain parameter `overlap' of a parallel run (domain.c:225,682; default 1 there, 0 here):
   the sweeps of relax_loop visit the cells along GO_SIDE_EXTERNAL sides first */
void go_domain_set_overlap (GoDomain * dom, int overlap)
{
  dom->mpi_order = overlap != 0;
}

GoField * go_field_new (GoDomain * dom, int component)
{
  GoField * f = calloc (1, sizeof (GoField));
  f->dom = dom;
  f->component = component;
  f->depth = dom->depth;
  for (int l = 0; l <= dom->depth; l++)
    f->lev[l] = calloc (dom->size[l], sizeof (double));
  return f;
}

void go_field_destroy (GoField * f)
{
  if (!f) return;
  for (int l = 0; l <= f->depth; l++)
    free (f->lev[l]);
  for (int d = 0; d < 6; d++)
    free (f->bcval[d]);
  free (f);
}

double * go_field_level (GoField * f, int level)
{
  return f->lev[level];
}

void go_field_set_bc (GoField * f, int d, int type, const double * val)
{
  GoDomain * dom = f->dom;
  f->bc[d] = type;
  free (f->bcval[d]);
  f->bcval[d] = NULL;
  if (val) {
    int n = dom->n[dom->depth];
    size_t m = dom->dim == 3 ? (size_t) n*n : (size_t) n;
    f->bcval[d] = malloc (m*sizeof (double));
    memcpy (f->bcval[d], val, m*sizeof (double));
  }
}

/* centre of cell (i,j,k) (1-based) of a unit box centred on the origin
 * (ftt_cell_pos, ftt.c:349-367: parent pos + coords*size/2, exact in binary) */
void go_cell_pos (const GoDomain * dom, int level, int i, int j, int k, double pos[3])
{
  double h = 1./dom->n[level];
  pos[0] = -0.5 + (i - 0.5)*h;
  pos[1] = -0.5 + (j - 0.5)*h;
  pos[2] = dom->dim == 3 ? -0.5 + (k - 0.5)*h : 0.;
}

/* Apply to v1 the boundary conditions of v on the ghost layer of `level`.
 * homogeneous = 0: gfs_domain_copy_bc (domain.c:846-867) with bc->bc;
 * homogeneous = 1: gfs_domain_homogeneous_bc (domain.c:945-965) with bc->homogeneous_bc.
 * Ghost-cell formulas: symmetry boundary.c:45-51, dirichlet :253-268, neumann :336-347,
 * periodic = value of the matching interior cell (boundary.c:1240-1258,1414-1442: every side
 * packs before any side unpacks, so only interior values are ever read). */
static void apply_bc (GoField * v, GoField * v1, int level, int homogeneous)
{
  GoDomain * dom = v1->dom;
  int n = dom->n[level], dim = dom->dim;
  double * a = v1->lev[level];
  double h = 1./n;
  int external = 0;
  int kmax = dim == 3 ? n : 1;

  for (int d = 0; d < 2*dim; d++) {
    int c = d/2;
    ptrdiff_t o = dom->off[level][d]; /* ghost = interior neighbour + o */
    if (dom->side[d] == GO_SIDE_EXTERNAL) { external = 1; continue; }
    for (int t2 = 1; t2 <= kmax; t2++)
      for (int t1 = 1; t1 <= n; t1++) {
	/* tangential coordinates in increasing axis order */
	int ijk[3] = { 0, 0, 0 };
	int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
	ijk[c] = (d & 1) ? 1 : n;
	ijk[ta] = t1;
	if (dim == 3) ijk[tb] = t2;
	size_t nb = go_index (dom, level, ijk[0], ijk[1], ijk[2]);
	size_t g = nb + o;
	if (dom->side[d] == GO_SIDE_PERIODIC) {
	  /* matching interior cell on the opposite side */
	  a[g] = a[nb - (ptrdiff_t) (n - 1)*o];
	  continue;
	}
	int type = v->bc[d];
	double val = 0.;
	if (!homogeneous && type != GO_BC_SYMMETRY && v->bcval[d]) {
	  assert (level == dom->depth); /* non-homogeneous values are held on leaves only */
	  val = v->bcval[d][(t1 - 1) + (size_t) (dim == 3 ? (t2 - 1)*n : 0)];
	}
	switch (type) {
	case GO_BC_SYMMETRY:
	  a[g] = (v->component == c) ? - a[nb] : a[nb];
	  break;
	case GO_BC_DIRICHLET:
	  a[g] = homogeneous ? - a[nb] : 2.*val - a[nb];
	  break;
	case GO_BC_NEUMANN:
	  a[g] = homogeneous ? a[nb] : a[nb] + val*h;
	  break;
	default: assert (0);
	}
      }
  }
  if (external && dom->exchange)
    (* dom->exchange) (dom->exchange_ctx, a, level, 0);
}

void go_bc (GoField * v, GoField * v1, int level)
{
  apply_bc (v, v1, level, 0);
}

void go_homogeneous_bc (GoField * ov, GoField * v, int level)
{
  apply_bc (v, ov, level, 1);
}

/* ---- binary cell data of a simulation file ---------------------------------------------------
 * ftt_cell_write_binary (ftt.c:1771-1799) with gfs_cell_write_binary (domain.c:3176-3207) as the
 * per-cell function, on the implicit uniform tree: pre-order, children n = 0..7 at
 * (x, y, z) = coords[n] (x:+ for bit 0, y:- for bit 1, z:- for bit 2, ftt.c:301-316); per cell
 * `guint flags' (child id | FTT_FLAG_LEAF = 1 << 4 on the deepest level), a double -1. (no
 * solid), one double per variable. */
static unsigned char * snapshot_cell (const GoDomain * dom, int nvars, GoField ** f,
				      unsigned char * p, int level, int i, int j, int k,
				      unsigned id)
{
  unsigned flags = id | (level == dom->depth ? 16u : 0u);
  memcpy (p, &flags, sizeof (unsigned)); p += sizeof (unsigned);
  double a = -1.;
  memcpy (p, &a, sizeof (double)); p += sizeof (double);
  size_t c = go_index (dom, level, i, j, dom->dim == 3 ? k : 0);
  for (int v = 0; v < nvars; v++) {
    a = f[v]->lev[level][c];
    memcpy (p, &a, sizeof (double)); p += sizeof (double);
  }
  if (level < dom->depth) {
    int nc = dom->dim == 3 ? 8 : 4;
    for (int n = 0; n < nc; n++) {
      /* child n of cell (i,j,k): 1-based coordinates on the finer level */
      int ci = 2*i - 1 + (n & 1);
      int cj = 2*j - ((n >> 1) & 1);
      int ck = dom->dim == 3 ? 2*k - ((n >> 2) & 1) : 0;
      p = snapshot_cell (dom, nvars, f, p, level + 1, ci, cj, ck, (unsigned) n);
    }
  }
  return p;
}

size_t go_snapshot_tree_bytes (const GoDomain * dom, int nvars)
{
  size_t cells = 0, c = 1;
  for (int l = 0; l <= dom->depth; l++) { cells += c; c *= dom->dim == 3 ? 8 : 4; }
  return cells*(sizeof (unsigned) + sizeof (double)*(1 + (size_t) nvars));
}

size_t go_snapshot_tree_write (const GoDomain * dom, int nvars, GoField ** f, unsigned char * buf)
{
  return (size_t) (snapshot_cell (dom, nvars, f, buf, 0, 1, 1, dom->dim == 3 ? 1 : 0, 0) - buf);
}
