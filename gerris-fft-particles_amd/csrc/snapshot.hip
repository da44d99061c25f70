// snapshot.hip -- the cell-tree data of a binary Gerris simulation file, to and from the per-level
// SoA arrays: what gfs_box_write / gfs_box_read put between the braces of a GfsBox when the domain
// has `binary = 1' (src/boundary.c:1819-1851,1853-2014):
//
//   ftt_cell_write_binary (src/ftt.c:1771-1799): pre-order over the tree, children n = 0..7 (n at
//     x:+ for bit 0, y:- for bit 1, z:- for bit 2, src/ftt.c:301-316); per cell `guint flags' =
//     child id (FTT_FLAG_ID) | FTT_FLAG_LEAF on the cells of the deepest level;
//   gfs_cell_write_binary (src/domain.c:3176-3207): a double -1. (no solid fractions), then one
//     double per variable of domain->variables_io.
//
// The reference walks 19 million pointer-linked cells for a 256^3 box; here the byte image is built
// (or taken apart) in HBM by one thread per cell: the offset of a cell's record follows from its
// level and coordinates alone -- on the way down from the root every step skips the parent's record
// and the subtrees of the n preceding siblings -- and the image moves over PCIe in one copy.
#include "gfship_internal.hpp"

namespace gfship {

#define SNAP_FLAG_LEAF 16u          /* FTT_FLAG_LEAF = 1 << 4, src/ftt.h:115 */
#define SNAP_MAXVARS 16

struct SnapArgs {
  int dim, depth, nvars;
  Layout lay[GFSHIP_MAXLEVEL + 1];
  double * v[SNAP_MAXVARS][GFSHIP_MAXLEVEL + 1];
  unsigned long long sub[GFSHIP_MAXLEVEL + 2];   // cells of the subtree of a cell of level l
  unsigned long long first[GFSHIP_MAXLEVEL + 2]; // index of the first thread of level l
  unsigned long long ncells;
  int rec;                                       // bytes per cell record
  unsigned char * image;
  unsigned * err;
};

__device__ __forceinline__ void put_u32 (unsigned char * p, unsigned v)
{
  *(unsigned *) p = v;      /* records are multiples of 4 bytes: 4-byte aligned */
}

__device__ __forceinline__ unsigned get_u32 (const unsigned char * p)
{
  return *(const unsigned *) p;
}

__device__ __forceinline__ void put_f64 (unsigned char * p, double v)
{
  const unsigned long long b = (unsigned long long) __double_as_longlong (v);
  put_u32 (p, (unsigned) b);
  put_u32 (p + 4, (unsigned) (b >> 32));
}

__device__ __forceinline__ double get_f64 (const unsigned char * p)
{
  const unsigned long long b = (unsigned long long) get_u32 (p) |
    ((unsigned long long) get_u32 (p + 4) << 32);
  return __longlong_as_double ((long long) b);
}

// one thread per cell of the tree, all levels
template <bool READ>
__global__ void __launch_bounds__(256)
snapshot_kernel (SnapArgs A)
{
  const unsigned long long q = (unsigned long long) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= A.ncells) return;
  int l = 0;
  while (l < A.depth && q >= A.first[l + 1]) l++;
  const unsigned long long c = q - A.first[l];
  const int n = 1 << l;
  // 0-based array coordinates: i grows with x, j with y, k with z
  const int i = (int) (c % n), j = (int) ((c / n) % n), k = A.dim == 3 ? (int) (c / ((unsigned long long) n*n)) : 0;
  // offset of the record, and the child id of the cell
  unsigned long long off = 0;
  unsigned id = 0;
  for (int d = 1; d <= l; d++) {
    const int sh = l - d;
    const unsigned bx = (i >> sh) & 1, by = (j >> sh) & 1, bz = (k >> sh) & 1;
    // child n: bit 0 set = +x half, bit 1 set = -y half, bit 2 set = -z half
    id = bx | ((by ^ 1u) << 1) | (A.dim == 3 ? ((bz ^ 1u) << 2) : 0u);
    off += 1 + id*A.sub[d];
  }
  unsigned char * p = A.image + off*A.rec;
  const unsigned flags = id | (l == A.depth ? SNAP_FLAG_LEAF : 0u);
  const long cell = A.lay[l].idx (i + 1, j + 1, A.dim == 3 ? k + 1 : 0);
  if (READ) {
    // cell_read_binary (src/ftt.c:1915-1945): the child id must match; the tree must be the
    // uniform one of this domain; no solid fractions (gfs_cell_read_binary, src/domain.c:3227-3236)
    const unsigned f = get_u32 (p);
    if ((f & 7u) != id) atomicOr (A.err, 1u);
    if (((f & SNAP_FLAG_LEAF) != 0) != (l == A.depth)) atomicOr (A.err, 2u);
    if (get_f64 (p + 4) != -1.) atomicOr (A.err, 4u);
    for (int v = 0; v < A.nvars; v++)
      A.v[v][l][cell] = get_f64 (p + 12 + 8*v);
  }
  else {
    put_u32 (p, flags);
    put_f64 (p + 4, -1.);
    for (int v = 0; v < A.nvars; v++)
      put_f64 (p + 12 + 8*v, A.v[v][l][cell]);
  }
}

static int snap_args (gfship_domain * dom, int nvars, const gfship_field * vars, SnapArgs * A)
{
  GFSHIP_CHECK (dom && vars, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (nvars >= 0 && nvars <= SNAP_MAXVARS, GFSHIP_EINVAL, "at most %d variables",
		SNAP_MAXVARS);
  A->dim = dom->dim; A->depth = dom->depth; A->nvars = nvars;
  const unsigned long long C = dom->dim == 3 ? 8 : 4;
  A->sub[dom->depth + 1] = 0;
  for (int l = dom->depth; l >= 0; l--)
    A->sub[l] = 1 + C*A->sub[l + 1];
  unsigned long long first = 0, cells = 1;
  for (int l = 0; l <= dom->depth; l++) {
    A->lay[l] = dom->lay[l];
    A->first[l] = first;
    first += cells;
    cells *= C;
  }
  A->first[dom->depth + 1] = first;
  A->ncells = first;
  A->rec = 4 + 8 + 8*nvars;
  for (int v = 0; v < nvars; v++) {
    Field * F = get_field (dom, vars[v]);
    if (!F) return GFSHIP_EINVAL;
    for (int l = 0; l <= dom->depth; l++)
      if (int r = coarse_flush (dom, F, l)) return r;
    for (int l = 0; l <= dom->depth; l++)
      A->v[v][l] = F->lev[l];
  }
  return GFSHIP_OK;
}

} // namespace gfship

using namespace gfship;

extern "C" {

size_t gfship_snapshot_tree_bytes (gfship_domain * dom, int nvars)
{
  if (!dom || nvars < 0) return 0;
  const unsigned long long C = dom->dim == 3 ? 8 : 4;
  unsigned long long cells = 0, c = 1;
  for (int l = 0; l <= dom->depth; l++) { cells += c; c *= C; }
  return (size_t) (cells*(4 + 8 + 8*(unsigned long long) nvars));
}

int gfship_snapshot_tree_write (gfship_domain * dom, int nvars, const gfship_field * vars,
				void * host_buf, size_t bytes)
{
  SnapArgs A;
  int r = snap_args (dom, nvars, vars, &A);
  if (r) return r;
  const size_t need = gfship_snapshot_tree_bytes (dom, nvars);
  GFSHIP_CHECK (host_buf && bytes >= need, GFSHIP_EINVAL, "the buffer must hold %zu bytes", need);
  unsigned char * image = nullptr;
  GFSHIP_HIP (hipMalloc ((void **) &image, need));
  A.image = image;
  A.err = nullptr;
  const unsigned blocks = (unsigned) ((A.ncells + 255)/256);
  hipLaunchKernelGGL (snapshot_kernel<false>, dim3 (blocks), dim3 (256), 0, dom->stream, A);
  hipError_t e = hipGetLastError ();
  if (e == hipSuccess)
    e = hipMemcpyAsync (host_buf, image, need, hipMemcpyDeviceToHost, dom->stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize (dom->stream);
  (void) hipFree (image);
  GFSHIP_HIP (e);
  return GFSHIP_OK;
}

int gfship_snapshot_tree_read (gfship_domain * dom, int nvars, const gfship_field * vars,
			       const void * host_buf, size_t bytes)
{
  SnapArgs A;
  int r = snap_args (dom, nvars, vars, &A);
  if (r) return r;
  if (dom->before_write && (r = dom->before_write (dom->before_write_ctx))) return r;
  const size_t need = gfship_snapshot_tree_bytes (dom, nvars);
  GFSHIP_CHECK (host_buf != nullptr, GFSHIP_EINVAL, "null buffer");
  GFSHIP_CHECK (bytes == need, GFSHIP_EINVAL,
		"the tree data of a uniform box refined to level %d with %d variables is %zu bytes, "
		"not %zu", dom->depth, nvars, need, bytes);
  unsigned char * image = nullptr;
  GFSHIP_HIP (hipMalloc ((void **) &image, need + 16));
  unsigned * err = (unsigned *) (image + ((need + 7) & ~(size_t) 7));
  A.image = image;
  A.err = err;
  hipError_t e = hipMemsetAsync (err, 0, sizeof (unsigned), dom->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync (image, host_buf, need, hipMemcpyHostToDevice, dom->stream);
  unsigned herr = 0;
  if (e == hipSuccess) {
    const unsigned blocks = (unsigned) ((A.ncells + 255)/256);
    hipLaunchKernelGGL (snapshot_kernel<true>, dim3 (blocks), dim3 (256), 0, dom->stream, A);
    e = hipGetLastError ();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync (&herr, err, sizeof (unsigned), hipMemcpyDeviceToHost, dom->stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize (dom->stream);
  (void) hipFree (image);
  GFSHIP_HIP (e);
  for (int v = 0; v < nvars; v++) {
    Field * F = get_field (dom, vars[v]);
    for (int l = 0; l <= dom->depth; l++)
      F->zero[l] = false;
    F->coarse_stale = false;
    F->coarse_valid = true;
  }
  GFSHIP_CHECK (!(herr & 1), GFSHIP_EINVAL,
		"FTT_CELL_ID (cell) != (flags & FTT_FLAG_ID): make sure the file has %d spatial "
		"dimensions", dom->dim);
  GFSHIP_CHECK (!(herr & 2), GFSHIP_EUNSUPPORTED,
		"the tree of the file is not the uniform tree refined to level %d", dom->depth);
  GFSHIP_CHECK (!(herr & 4), GFSHIP_EUNSUPPORTED, "the file has solid fractions (mixed cells)");
  return GFSHIP_OK;
}

} // extern "C"
