/* go_aos.c -- oracle, CPU baseline variant: the relax sweep (src/poisson.c:507-530) on a
 * reference-like cell tree instead of the oracle's flat arrays.  TEST INFRASTRUCTURE ONLY.
 *
 * "Where the time goes today" (SURVEY.md 8d): the reference stores every cell as a record reached
 * through pointers -- FttCell { flags, data, parent, children } and FttOct { level, parent,
 * neighbours of the parent, position, cell[8] } (src/ftt.h:134-159), the per-cell state
 * GfsStateVector { f[6]{un, v}, solid, vars[] } in one block per oct (src/fluid.h:39-52,
 * src/domain.c:2939) -- finds neighbours with ftt_cell_neighbor (src/ftt.h:500-545: sibling of the
 * same oct, or the parent's cached neighbour and then its child) and visits the leaves by a
 * recursive pre-order traversal calling a function pointer per cell (src/ftt.c:837-852).  This file
 * restates exactly that access pattern for a full uniform periodic octree so that bench.py can time
 * it next to the flat-array oracle; tests check that it reproduces go_relax bit for bit. */
#include <stdlib.h>
#include <string.h>
#include "gfs_oracle.h"

#define NVARS 12                       /* P Pmac U V W + gradients ... of a default GfsSimulation */
enum { VAR_U = 0, VAR_RHS = 1, VAR_DIA = 2, VAR_GHOST = 3 };

typedef struct AosCell AosCell;
typedef struct AosOct AosOct;
typedef struct { struct { double un, v; } f[6]; void * solid; double vars[NVARS]; } AosState;
struct AosCell { unsigned flags; AosState * data; AosOct * parent; AosOct * children; };
struct AosOct { unsigned level; AosCell * parent; AosCell * neighbors[6]; double pos[3]; AosCell cell[8]; };

typedef struct { AosCell root; AosCell * root_neighbors[6]; int depth; size_t ncells; } AosTree;

/* neighbour of child n in direction d: >= 0 sibling, < 0 child (-v - 1) of the parent's neighbour
   (the table of ftt_cell_neighbor_not_cached for the child numbering of src/ftt.c:301-316:
   bit 0 = +x, bit 1 = -y, bit 2 = -z) */
static int neighbor_index (int d, int n)
{
  int axis = d/2, plus = !(d & 1);
  int bit = 1 << axis;
  int towards_plus_side = axis == 0 ? (n & bit) != 0 : (n & bit) == 0;  /* child sits on the + side */
  int sibling = n ^ bit;
  if (plus ? !towards_plus_side : towards_plus_side)
    return sibling;
  return - sibling - 1;
}

static AosCell * cell_neighbor (AosTree * t, AosCell * cell, int d)
{
  if (!cell->parent)
    return t->root_neighbors[d];
  int n = (int) (cell - cell->parent->cell);
  int nn = neighbor_index (d, n);
  if (nn >= 0)
    return &cell->parent->cell[nn];
  AosCell * c = cell->parent->neighbors[d];
  if (c && c->children)
    c = &c->children->cell[- nn - 1];
  return c;
}

static void refine (AosTree * t, AosCell * cell, int level)
{
  if (level == t->depth) return;
  AosOct * oct = calloc (1, sizeof (AosOct));
  oct->level = level;
  oct->parent = cell;
  cell->children = oct;
  AosState * block = calloc (8, sizeof (AosState));     /* one block per oct, like the reference */
  for (int n = 0; n < 8; n++) {
    oct->cell[n].parent = oct;
    oct->cell[n].data = block + n;
    t->ncells++;
  }
}

/* level by level, so that the parent's neighbours exist when an oct caches them */
static void build_level (AosTree * t, AosCell * cell, int level, int target)
{
  if (level == target) {
    refine (t, cell, level);
    if (cell->children)
      for (int d = 0; d < 6; d++)
	cell->children->neighbors[d] = cell_neighbor (t, cell, d);
    return;
  }
  for (int n = 0; n < 8; n++)
    build_level (t, &cell->children->cell[n], level + 1, target);
}

void * go_aos_new (int depth)
{
  AosTree * t = calloc (1, sizeof (AosTree));
  t->depth = depth;
  t->root.data = calloc (1, sizeof (AosState));
  for (int d = 0; d < 6; d++) t->root_neighbors[d] = &t->root;       /* periodic box */
  for (int l = 0; l < depth; l++)
    build_level (t, &t->root, 0, l);
  return t;
}

static void free_cell (AosCell * cell)
{
  if (!cell->children) return;
  for (int n = 0; n < 8; n++) free_cell (&cell->children->cell[n]);
  free (cell->children->cell[0].data);
  free (cell->children);
}

void go_aos_destroy (void * tree)
{
  AosTree * t = tree;
  free_cell (&t->root);
  free (t->root.data);
  free (t);
}

typedef void (* CellFunc) (AosTree * t, AosCell * cell, int i, int j, int k, void * data);

/* pre-order over the leaves with the grid coordinates of the cell (1-based like the oracle) */
static void traverse_leaves (AosTree * t, AosCell * cell, int level, int i, int j, int k,
			     CellFunc func, void * data)
{
  if (!cell->children) {
    (* func) (t, cell, i + 1, j + 1, k + 1, data);
    return;
  }
  for (int n = 0; n < 8; n++)
    traverse_leaves (t, &cell->children->cell[n], level + 1,
		     2*i + ((n & 1) ? 1 : 0), 2*j + ((n & 2) ? 0 : 1), 2*k + ((n & 4) ? 0 : 1),
		     func, data);
}

typedef struct { const GoDomain * dom; double * a[3]; int to_tree; } CopyData;

static void copy_cell (AosTree * t, AosCell * cell, int i, int j, int k, void * data)
{
  CopyData * c = data;
  size_t idx = go_index (c->dom, c->dom->depth, i, j, k);
  for (int v = 0; v < 3; v++)
    if (c->a[v]) {
      if (c->to_tree) cell->data->vars[v] = c->a[v][idx];
      else c->a[v][idx] = cell->data->vars[v];
    }
}

/* leaf values of u, rhs, dia (flat oracle arrays with ghosts) into / out of the tree */
void go_aos_load (void * tree, const GoDomain * dom, double * u, double * rhs, double * dia)
{
  CopyData c = { dom, { u, rhs, dia }, 1 };
  traverse_leaves (tree, &((AosTree *) tree)->root, 0, 0, 0, 0, copy_cell, &c);
  /* poisson coefficients of a uniform periodic box: every face weight is 1 (poisson.c:756-901) */
}

void go_aos_store (void * tree, const GoDomain * dom, double * u)
{
  CopyData c = { dom, { u, NULL, NULL }, 0 };
  traverse_leaves (tree, &((AosTree *) tree)->root, 0, 0, 0, 0, copy_cell, &c);
}

/* The reference's periodic sides are boundary cells filled by gfs_domain_homogeneous_bc before the
   sweep (src/boundary.c:1240-1451): a cell next to a side reads the value its image had BEFORE the
   sweep.  Here the box is its own neighbour, so that value is kept per cell by a pass standing in for
   the BC traversal. */
static void snapshot_cell (AosTree * t, AosCell * cell, int i, int j, int k, void * data)
{
  cell->data->vars[VAR_GHOST] = cell->data->vars[VAR_U];
}

/* relax, src/poisson.c:507-530 */
static void relax_cell (AosTree * t, AosCell * cell, int i, int j, int k, void * data)
{
  AosState * s = cell->data;
  const int n = 1 << t->depth;
  const int ijk[3] = { i, j, k };
  double a = s->vars[VAR_DIA], b = 0.;
  for (int d = 0; d < 6; d++) {
    AosCell * nb = cell_neighbor (t, cell, d);
    if (nb) {
      double w = s->f[d].v;
      int across = (d & 1) ? ijk[d/2] == 1 : ijk[d/2] == n;     /* through a side of the box */
      a += w;
      b += w*nb->data->vars[across ? VAR_GHOST : VAR_U];
    }
  }
  if (a != 0.)
    s->vars[VAR_U] = (b - s->vars[VAR_RHS])/a;
  else
    s->vars[VAR_U] = 0.;
}

static void unit_weights (AosTree * t, AosCell * cell, int i, int j, int k, void * data)
{
  for (int d = 0; d < 6; d++) cell->data->f[d].v = 1.;
}

void go_aos_relax (void * tree, int nsweeps)
{
  AosTree * t = tree;
  traverse_leaves (t, &t->root, 0, 0, 0, 0, unit_weights, NULL);
  for (int s = 0; s < nsweeps; s++) {
    traverse_leaves (t, &t->root, 0, 0, 0, 0, snapshot_cell, NULL);
    traverse_leaves (t, &t->root, 0, 0, 0, 0, relax_cell, NULL);
  }
}

size_t go_aos_bytes_per_cell (void) { return sizeof (AosCell) + sizeof (AosState) + sizeof (AosOct)/8; }
