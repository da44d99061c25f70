// tree.hpp -- the coarse-fine stencils of a statically refined quadtree / octree.
//
// The reference keeps a tree of FttOct records and reaches a neighbour, a parent or the children
// of a cell through pointers (src/ftt.h:134-159,518-573).  Here level l of the tree is a dense
// (n + 2)^dim array, n = 2^l, with one ghost layer (the ghost trees of the periodic sides); all levels
// of a variable sit behind each other in one allocation, and a byte per cell says whether the cell
// exists and whether it is a leaf.  Neighbour, parent and children are index arithmetic.
//
// Everything in this file is __host__ __device__ and templated on how a value is read: the kernels
// (tree.hip) read device arrays; the host instantiates the same stencil code with a reader that
// records WHICH cells are read, and derives from that the dependency levels of an exact-order sweep
// (the reference relaxes the cells of a level, coarser leaves included, in tree order:
// src/poisson.c:604-632, src/ftt.c:689-926).
//
// Restated with the fine / coarse branches (unit face weights, no solid fractions; FTT_2D and FTT_3D):
//   average_neighbor_value src/fluid.c:64-93        interpolate_1D1 :178-197, interpolate_2D1 :214-245
//   gradient_fine_coarse   :283-309                 gfs_neighbor_value :364-396
//   gfs_center_gradient    :434-475                 gfs_face_gradient :778-829
//   face_weighted_gradient :833-893 (w = 1: the same numbers as gfs_face_gradient)
//   gfs_face_interpolated_value :2186-2198
#pragma once
#include <hip/hip_runtime.h>
#include "gfship.h"

namespace gfship { namespace tree {

enum { NONE = 0, LEAF = 1, NODE = 2 };

struct Cell { int l, q; };              // q < 0: no such cell

struct Face { Cell cell, neighbor; int d; };   // FttCellFace

struct Topo {
  int dim;                              // 2 or 3
  int depth;
  int off[GFSHIP_MAXLEVEL + 2];         // first cell of each level in the concatenated arrays
  const unsigned char * flag;
  // the tree is static: once it is built the answers of neighbor (), child () and id () are read
  // from tables (one load instead of a chain of index divisions and flag loads -- the sweeps of a
  // tree are bound by the latency of such chains); nullptr while the tree is being built
  const int * nbtab = nullptr;          // [cell*nd + d]: q of the neighbour (>= 0 same level, <= -2: -q - 2 one level up, -1 none)
  const int * child0 = nullptr;         // [cell]: q of the child (2i - 1, 2j, 2k) on the next level
  const unsigned char * cmask = nullptr;// [cell]: bit c = child c exists
  const unsigned char * idtab = nullptr;// [cell]: FTT_CELL_ID | 8*interior

  __host__ __device__ inline int nd () const { return 2*dim; }           // FTT_NEIGHBORS
  __host__ __device__ inline int nc () const { return 1 << dim; }        // FTT_CELLS
  __host__ __device__ inline int ncd () const { return 1 << (dim - 1); } // FTT_CELLS_DIRECTION
  __host__ __device__ inline int n (int l) const { return 1 << l; }
  __host__ __device__ inline int r (int l) const { return (1 << l) + 2; }
  __host__ __device__ inline int lsize (int l) const { return dim == 3 ? r (l)*r (l)*r (l) : r (l)*r (l); }
  __host__ __device__ inline int gi (Cell c) const { return off[c.l] + c.q; }
  __host__ __device__ inline int ci (Cell c) const { return c.q % r (c.l); }
  __host__ __device__ inline int cj (Cell c) const { return (c.q / r (c.l)) % r (c.l); }
  __host__ __device__ inline int ck (Cell c) const { return dim == 3 ? c.q / (r (c.l)*r (c.l)) : 1; }
  __host__ __device__ inline bool leaf (Cell c) const { return flag[gi (c)] == LEAF; }
  __host__ __device__ inline Cell make (int l, int i, int j, int k) const {
    Cell c = { l, -1 };
    if (l < 0 || l > depth || i < 0 || j < 0 || i > n (l) + 1 || j > n (l) + 1)
      return c;
    if (dim == 3 && (k < 0 || k > n (l) + 1))
      return c;
    int q = i + r (l)*(j + (dim == 3 ? r (l)*k : 0));
    if (flag[off[l] + q] != NONE)
      c.q = q;
    return c;
  }
  // FTT_CELL_ID (src/ftt.c:301-316): bit 0 = +x, bit 1 = -y, bit 2 = -z
  __host__ __device__ inline int id (Cell c) const {
    if (idtab) return idtab[gi (c)] & 7;
    return ((ci (c) + 1) & 1) + 2*(cj (c) & 1) + (dim == 3 ? 4*(ck (c) & 1) : 0);
  }
  // ftt_cell_neighbor, src/ftt.h:518-573
  __host__ __device__ inline Cell neighbor (Cell c, int d) const {
    if (nbtab) {
      const int v = nbtab[gi (c)*nd () + d];
      Cell nb = { c.l, v };
      if (v <= -2) { nb.l = c.l - 1; nb.q = - v - 2; }
      return nb;
    }
    const int i = ci (c) + (d == 0) - (d == 1), j = cj (c) + (d == 2) - (d == 3),
      k = ck (c) + (d == 4) - (d == 5);
    Cell nb = make (c.l, i, j, k);
    if (nb.q >= 0 || c.l == 0 || i < 0 || j < 0 || k < 0 ||
	i > n (c.l) + 1 || j > n (c.l) + 1 || k > n (c.l) + 1)
      return nb;
    return make (c.l - 1, (i + 1)/2, (j + 1)/2, (k + 1)/2);
  }
  __host__ __device__ inline Cell child (Cell c, int k) const {
    if (cmask) {
      const int g = gi (c);
      Cell ch = { c.l + 1, -1 };
      if ((cmask[g] >> k) & 1) {
	const int rr = r (c.l + 1);
	ch.q = child0[g] + (k & 1) - rr*((k >> 1) & 1) - (dim == 3 ? rr*rr*((k >> 2) & 1) : 0);
      }
      return ch;
    }
    return make (c.l + 1, 2*ci (c) - 1 + (k & 1), 2*cj (c) - ((k >> 1) & 1), 2*ck (c) - ((k >> 2) & 1));
  }
  // ftt_cell_children_direction, src/ftt.h:321-355: child i (of ncd ()) on the side d of the cell
  __host__ __device__ inline Cell child_direction (Cell c, int d, int i) const {
    // the child ids with the bit of the axis of d fixed (set for d = 0, 3, 5), the other bits = i
    const int a = d >> 1;
    const int fixed = (d == 0 || d == 3 || d == 5) ? 1 : 0;
    int idx;
    if (dim == 2)
      idx = a == 0 ? (fixed | (i << 1)) : (i | (fixed << 1));
    else
      idx = a == 0 ? (fixed | (i << 1)) :
	a == 1 ? ((i & 1) | (fixed << 1) | ((i >> 1) << 2)) : (i | (fixed << 2));
    return child (c, idx);
  }
  // ftt_cell_child_corner, src/ftt.h:366-425: the child in the corner of the given directions
  __host__ __device__ inline Cell child_corner (Cell c, int d0, int d1, int d2) const {
    int idx = 0;
    if (d0 == 0 || d1 == 0 || d2 == 0) idx |= 1;
    if (d0 == 3 || d1 == 3 || d2 == 3) idx |= 2;
    if (d0 == 5 || d1 == 5 || d2 == 5) idx |= 4;
    return child (c, idx);
  }
  __host__ __device__ inline bool interior (Cell c) const {
    if (idtab) return (idtab[gi (c)] & 8) != 0;
    const int i = ci (c), j = cj (c), k = ck (c);
    return i >= 1 && j >= 1 && k >= 1 && i <= n (c.l) && j <= n (c.l) && k <= n (c.l);
  }
  __host__ __device__ inline double size (Cell c) const { return 1./(1 << c.l); }
};

__host__ __device__ inline bool exists (Cell c) { return c.q >= 0; }
__host__ __device__ inline bool fine_coarse (const Face & f) { return f.neighbor.l < f.cell.l; }

// src/fluid.c:200-213: the directions, seen from the coarse neighbour, in which the fine cell sits:
// 2-D {{-1,2,-1,3},{2,-1,3,-1},{1,0,-1,-1},{-1,-1,1,0}}; 3-D perpendicular[d][id][2]: for an x face
// (y, z), for a y face (z, x), for a z face (x, y) -- in that order
__host__ __device__ inline int axis_direction (int axis, int id)
{
  return axis == 0 ? ((id & 1) ? 0 : 1) : axis == 1 ? ((id & 2) ? 3 : 2) : ((id & 4) ? 5 : 4);
}
__host__ __device__ inline int perpendicular (int d, int id)                 /* FTT_2D */
{
  return axis_direction (d < 2 ? 1 : 0, id);
}
__host__ __device__ inline int perpendicular3 (int d, int id, int which)     /* FTT_3D */
{
  const int a = d >> 1;
  return axis_direction ((a + 1 + which) % 3, id);
}

struct Grad2 { double a, b; };          // GfsGradient: v = a*v(cell) + b
struct Grad3 { double a, b, c; };

template <class V>
__host__ __device__ inline double average_neighbor_value (const Topo & T, const Face & face, V & v, double & x)
{
  if (T.leaf (face.neighbor))
    return v (T, face.neighbor);
  double av = 0., a = 0.;
  for (int i = 0; i < T.ncd (); i++) {
    const Cell ch = T.child_direction (face.neighbor, face.d ^ 1, i);
    if (exists (ch)) {
      a += 1.;
      av += 1.*v (T, ch);
    }
  }
  if (a > 0.) {
    x = 3./4.;
    return av/a;
  }
  return v (T, face.cell);
}

template <class V>
__host__ __device__ inline Grad2 interpolate_1D1 (const Topo & T, Cell cell, int d, double x, V & v)
{
  Grad2 p = { 1., 0. };
  const Face f = { cell, T.neighbor (cell, d), d };
  if (exists (f.neighbor)) {
    double x2 = 1.;
    const double p2 = average_neighbor_value (T, f, v, x2);
    const double a2 = x/x2;
    p.b += a2*p2;
    p.a -= a2;
  }
  return p;
}

template <class V>
__host__ __device__ inline Grad2 interpolate_2D1 (const Topo & T, Cell cell, int d1, int d2, double x,
						  double y, V & v)
{
  Grad2 p = { 1., 0. };
  const Face f1 = { cell, T.neighbor (cell, d1), d1 };
  if (exists (f1.neighbor)) {
    double y1 = 1.;
    const double p1 = average_neighbor_value (T, f1, v, y1);
    const double a1 = y/y1;
    p.b += a1*p1;
    p.a -= a1;
  }
  const Face f2 = { cell, T.neighbor (cell, d2), d2 };
  if (exists (f2.neighbor)) {
    double x2 = 1.;
    const double p2 = average_neighbor_value (T, f2, v, x2);
    const double a2 = x/x2;
    p.b += a2*p2;
    p.a -= a2;
  }
  return p;
}

// the interpolation in the coarse neighbour of a fine-coarse face towards the fine cell
template <class V>
__host__ __device__ inline Grad2 interpolate_coarse (const Topo & T, const Face & face, V & v)
{
  const int id = T.id (face.cell);
  if (T.dim == 2)
    return interpolate_1D1 (T, face.neighbor, perpendicular (face.d, id), 1./4., v);
  return interpolate_2D1 (T, face.neighbor, perpendicular3 (face.d, id, 0), perpendicular3 (face.d, id, 1),
			  1./4., 1./4., v);
}

template <class V>
__host__ __device__ inline Grad3 gradient_fine_coarse (const Topo & T, const Face & face, V & v)
{
  const Grad2 p = interpolate_coarse (T, face, v);
  Grad3 g;
  g.a = 2./3.;
  g.b = 2.*p.a/3.;
  g.c = 2.*p.b/3.;
  return g;
}

template <class V>
__host__ __device__ inline double neighbor_value (const Topo & T, const Face & face, V & v, double & x)
{
  if (face.neighbor.l == face.cell.l)
    return average_neighbor_value (T, face, v, x);
  const Grad2 vc = interpolate_coarse (T, face, v);
  x = 3./2.;
  return vc.a*v (T, face.neighbor) + vc.b;
}

template <class V>
__host__ __device__ inline double center_gradient (const Topo & T, Cell cell, int c, V & v)
{
  const int d = 2*c;
  const Face f1 = { cell, T.neighbor (cell, d ^ 1), d ^ 1 };
  const Face f2 = { cell, T.neighbor (cell, d), d };
  const double v0 = v (T, cell);
  if (exists (f1.neighbor)) {
    double x1 = 1.;
    const double v1 = neighbor_value (T, f1, v, x1);
    if (exists (f2.neighbor)) {
      double x2 = 1.;
      const double v2 = neighbor_value (T, f2, v, x2);
      return (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
    }
    return (v0 - v1)/x1;
  }
  if (exists (f2.neighbor)) {
    double x2 = 1.;
    return (neighbor_value (T, f2, v, x2) - v0)/x2;
  }
  return 0.;
}

// gfs_center_van_leer_gradient, src/fluid.c:522-561
template <class V>
__host__ __device__ inline double van_leer_gradient (const Topo & T, Cell cell, int c, V & v)
{
  const int d = 2*c;
  const Face f1 = { cell, T.neighbor (cell, d ^ 1), d ^ 1 };
  if (exists (f1.neighbor)) {
    const Face f2 = { cell, T.neighbor (cell, d), d };
    if (exists (f2.neighbor)) {
      double x1 = 1., x2 = 1.;
      const double v0 = v (T, cell);
      const double v1 = neighbor_value (T, f1, v, x1);
      const double v2 = neighbor_value (T, f2, v, x2);
      double s1 = 2.*(v0 - v1);
      const double s2 = 2.*(v2 - v0);
      if (s1*s2 <= 0.)
	return 0.;
      const double s0 = (x1*x1*(v2 - v0) + x2*x2*(v0 - v1))/(x1*x2*(x2 + x1));
      if (fabs (s2) < fabs (s1))
	s1 = s2;
      if (fabs (s0) < fabs (s1))
	return s0;
      return s1;
    }
  }
  return 0.;
}

// gfs_face_gradient; with unit weights also face_weighted_gradient (dimension 2)
template <class V>
__host__ __device__ inline Grad2 face_gradient (const Topo & T, const Face & face, V & v, int max_level)
{
  Grad2 g = { 0., 0. };
  if (!exists (face.neighbor))
    return g;
  const int level = face.cell.l;
  if (face.neighbor.l < level) {
    const Grad3 gcf = gradient_fine_coarse (T, face, v);
    g.a = gcf.a;
    g.b = gcf.b*v (T, face.neighbor) + gcf.c;
  }
  else if (level == max_level || T.leaf (face.neighbor)) {
    g.a = 1.;
    g.b = v (T, face.neighbor);
  }
  else {
    Face f;
    f.d = face.d ^ 1;
    f.neighbor = face.cell;
    const int n = T.ncd ();
    for (int i = 0; i < n; i++) {
      f.cell = T.child_direction (face.neighbor, f.d, i);
      if (exists (f.cell)) {
	const Grad3 gcf = gradient_fine_coarse (T, f, v);
	g.a += 1.*gcf.b;
	g.b += 1.*(gcf.a*v (T, f.cell) - gcf.c);
      }
    }
    if (T.dim > 2) {     /* n/2. = 1 in 2-D */
      g.a /= n/2.;
      g.b /= n/2.;
    }
  }
  return g;
}

template <class V>
__host__ __device__ inline double face_interpolated_value (const Topo & T, const Face & face, V & v)
{
  double x1 = 1.;
  if (exists (face.neighbor)) {
    const double v1 = neighbor_value (T, face, v, x1);
    return ((x1 - 0.5)*v (T, face.cell) + 0.5*v1)/x1;
  }
  return v (T, face.cell);
}

// gfs_face_interpolated_value_generic, src/fluid.c:2200-2221
template <class V>
__host__ __device__ inline double face_interpolated_value_generic (const Topo & T, const Face & face, V & v)
{
  if (!exists (face.neighbor) || T.leaf (face.neighbor) || face.neighbor.l < face.cell.l)
    return face_interpolated_value (T, face, v);
  Face f;
  f.neighbor = face.cell;
  f.d = face.d ^ 1;
  const int n = T.ncd ();
  double avg = 0.;
  for (int i = 0; i < n; i++) {
    f.cell = T.child_direction (face.neighbor, f.d, i);
    if (exists (f.cell))
      avg += face_interpolated_value (T, f, v)*1.;
  }
  return avg == 0. ? 0. : avg/(1.*n);
}

// face_weighted_gradient (src/fluid.c:833-893) / gfs_face_cm_weighted_gradient (:1300-1400, no metric)
// with the SAME weight w on every face: the face coefficients of gfs_diffusion_coefficients with a
// constant D on a quadtree (diffusion_coef gives every leaf face w, the coarse side of a fine-coarse
// face w/2 + w/2 and face_coeff_from_below (w + w)/2, all exactly w; in 3-D the sums of four quarters
// may round: not used there)
template <class V>
__host__ __device__ inline Grad2 face_gradient_w (const Topo & T, const Face & face, V & v, int max_level, double w)
{
  Grad2 g = { 0., 0. };
  if (!exists (face.neighbor) || w == 0.)
    return g;
  const int level = face.cell.l;
  if (face.neighbor.l < level) {
    const Grad3 gcf = gradient_fine_coarse (T, face, v);
    g.a = w*gcf.a;
    g.b = w*(gcf.b*v (T, face.neighbor) + gcf.c);
  }
  else if (level == max_level || T.leaf (face.neighbor)) {
    g.a = w;
    g.b = w*v (T, face.neighbor);
  }
  else {
    Face f;
    f.d = face.d ^ 1;
    f.neighbor = face.cell;
    const int n = T.ncd ();
    for (int i = 0; i < n; i++) {
      f.cell = T.child_direction (face.neighbor, f.d, i);
      if (exists (f.cell)) {
	const Grad3 gcf = gradient_fine_coarse (T, f, v);
	g.a += w*gcf.b;
	g.b += w*(gcf.a*v (T, f.cell) - gcf.c);
      }
    }
    if (T.dim > 2) {
      g.a /= n/2.;
      g.b /= n/2.;
    }
  }
  return g;
}

// diffusion_relax, src/poisson.c:1455-1484 (rhoc = 1): the new value of u at `cell'
template <class V>
__host__ __device__ inline double diffusion_relax_cell (const Topo & T, Cell cell, V & u, double res, double w,
							int max_level)
{
  Grad2 g = { 0., 0. };
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < T.nd (); f.d++) {
    f.neighbor = T.neighbor (cell, f.d);
    const Grad2 ng = face_gradient_w (T, f, u, max_level, w);
    g.a += ng.a;
    g.b += ng.b;
  }
  const double h = T.size (cell);
  const double a = 1.*h*h;
  g.a = 1. + g.a/a;
  return (g.b/a + res)/g.a;
}

// diffusion_residual, src/poisson.c:1519-1556 (rhoc = 1)
template <class V>
__host__ __device__ inline double diffusion_residual_cell (const Topo & T, Cell cell, V & u, double rhs, double w)
{
  Grad2 g = { 0., 0. };
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < T.nd (); f.d++) {
    f.neighbor = T.neighbor (cell, f.d);
    const Grad2 ng = face_gradient_w (T, f, u, -1, w);
    g.a += ng.a;
    g.b += ng.b;
  }
  const double h = T.size (cell);
  double a = 1.;
  a *= h*h;
  g.a = 1. + g.a/a;
  g.b = rhs + g.b/a;
  return g.b - g.a*u (T, cell);
}

// diffusion_rhs, src/poisson.c:1392-1421 (rhoc = 1): what is added to rhs
template <class V>
__host__ __device__ inline double diffusion_rhs_cell (const Topo & T, Cell cell, V & v, double w, double pbeta)
{
  double f = 0.;
  const double h = T.size (cell), value = v (T, cell);
  Face face;
  face.cell = cell;
  for (face.d = 0; face.d < T.nd (); face.d++) {
    face.neighbor = T.neighbor (cell, face.d);
    const Grad2 g = face_gradient_w (T, face, v, -1, w);
    f += g.b - g.a*value;
  }
  return pbeta*f/(h*h*1.);
}

// source_diffusion_value, src/source.c:1105-1144 (phi = v, constant D, alpha = NULL)
template <class V>
__host__ __device__ inline double source_diffusion_value (const Topo & T, Cell cell, V & v, double D)
{
  Grad2 g = { 0., 0. };
  const double v0 = v (T, cell);
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < T.nd (); f.d++) {
    f.neighbor = T.neighbor (cell, f.d);
    if (exists (f.neighbor)) {
      const Grad2 e = face_gradient (T, f, v, -1);
      g.a += D*e.a;
      g.b += D*e.b;
    }
  }
  const double h = T.size (cell);
  return 1.*(g.b - g.a*v0)/(h*h);
}

// relax2D (src/poisson.c:532-557) / relax (:507-530, no omega), dia = 0, unit weights: the new value of u at `cell'
template <class V>
__host__ __device__ inline double relax_cell (const Topo & T, Cell cell, V & u, double rhs, double omega,
					      int max_level)
{
  Grad2 g = { 0., 0. };
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < T.nd (); f.d++) {
    f.neighbor = T.neighbor (cell, f.d);
    if (exists (f.neighbor)) {
      const Grad2 ng = face_gradient (T, f, u, max_level);
      g.a += ng.a;
      g.b += ng.b;
    }
  }
  if (g.a != 0.) {
    if (T.dim == 2)
      return (1. - omega)*u (T, cell) + omega*(g.b - rhs)/g.a;
    (void) u (T, cell);      /* the recording reader must see the cell itself in 3-D too */
    return (g.b - rhs)/g.a;
  }
  return 0.;
}

// residual_set2D, src/poisson.c:657-678
template <class V>
__host__ __device__ inline double residual_cell (const Topo & T, Cell cell, V & u, double rhs)
{
  Grad2 g = { 0., 0. };
  Face f;
  f.cell = cell;
  for (f.d = 0; f.d < T.nd (); f.d++) {
    f.neighbor = T.neighbor (cell, f.d);
    if (exists (f.neighbor)) {
      const Grad2 ng = face_gradient (T, f, u, -1);
      g.a += ng.a;
      g.b += ng.b;
    }
  }
  return rhs - (g.b - u (T, cell)*g.a);
}

} } // namespace gfship::tree
