// bc.hip -- K17: ghost-layer fill (symmetry, Dirichlet, Neumann, periodic).
// Replaces box_bc / box_homogeneous_bc + gfs_boundary_send/receive of src/domain.c:723-965 and
// the periodic pack/memcpy/unpack of src/boundary.c:1240-1451: on one box every side reads
// interior cells only (all sides pack before any side unpacks), so all 2*dim faces are filled
// by one launch.
#include "gfship_internal.hpp"

namespace gfship {

__device__ __forceinline__ double ghost_value_bc (int type, int component, int c, double nb,
						  int homogeneous, double val, double h)
{
  switch (type) {
  case GFSHIP_BC_DIRICHLET:        /* src/boundary.c:253-268 */
    return homogeneous ? - nb : 2.*val - nb;
  case GFSHIP_BC_NEUMANN:          /* src/boundary.c:336-347 */
    return homogeneous ? nb : nb + val*h;
  default:                         /* symmetry, src/boundary.c:45-51 */
    return component == c ? - nb : nb;
  }
}

__global__ void __launch_bounds__(256)
bc_kernel (Layout L, BcDesc bc, double * __restrict__ a)
{
  const int n = L.n;
  const int nface = L.dim == 3 ? n*n : n;
  int f = blockIdx.x*blockDim.x + threadIdx.x;
  int d = blockIdx.y;
  if (f >= nface) return;
  if (bc.side[d] == GFSHIP_SIDE_EXTERNAL) return;
  int c = d/2;
  int t1 = f % n + 1, t2 = L.dim == 3 ? f / n + 1 : 0;
  int ijk[3] = { 0, 0, 0 };
  int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
  ijk[c] = (d & 1) ? 1 : n;
  ijk[ta] = t1;
  if (L.dim == 3) ijk[tb] = t2;
  long o = c == 0 ? 1 : c == 1 ? L.sy : L.sz;
  if (d & 1) o = - o;
  long nb = L.idx (ijk[0], ijk[1], ijk[2]);
  double v;
  if (bc.side[d] == GFSHIP_SIDE_PERIODIC)
    v = a[nb - (long) (n - 1)*o];
  else {
    double val = (!bc.homogeneous && bc.val[d]) ? bc.val[d][f] : 0.;
    v = ghost_value_bc (bc.type[d], bc.component, c, a[nb], bc.homogeneous, val, 1./n);
  }
  a[nb + o] = v;
}

static int launch_bc_kernel (gfship_domain * dom, Field * v, Field * v1, int level, int homogeneous);

int launch_bc (gfship_domain * dom, Field * v, Field * v1, int level, int homogeneous)
{
  int r = launch_bc_kernel (dom, v, v1, level, homogeneous);
  if (r) return r;
  /* GfsBoundaryMpi sides: ghost layer from the neighbour boxes */
  return call_exchange (dom, v1->lev[level], level, 0);
}

// the two halves of a BC application around the bulk of a sweep (overlap = 1): the exchange of the
// MPI sides starts after the cells along them have been swept; the local sides are filled and the
// arrivals unpacked after the bulk.  With the hooks of the caller instead of the library's
// communicator nothing can run beside the sweep: the whole exchange happens in the second half
// (same values: the layers do not change after the first half).
int bc_mpi_begin (gfship_domain * dom, Field * v1, int level)
{
  if (!dom->has_external || !dom->comm) return GFSHIP_OK;
  return comm_exchange_begin (dom, v1->lev[level], level);
}

int bc_mpi_end (gfship_domain * dom, Field * v, Field * v1, int level, int homogeneous)
{
  int r = launch_bc_kernel (dom, v, v1, level, homogeneous);
  if (r) return r;
  if (!dom->has_external) return GFSHIP_OK;
  if (dom->comm) return comm_exchange_end (dom, v1->lev[level], level);
  return call_exchange (dom, v1->lev[level], level, 0);
}

static int launch_bc_kernel (gfship_domain * dom, Field * v, Field * v1, int level, int homogeneous)
{
  const Layout & L = dom->lay[level];
  {
    /* a box whose sides all face other boxes: every ghost cell comes with the exchange */
    bool local = false;
    for (int d = 0; d < 2*dom->dim; d++)
      if (dom->side[d] != GFSHIP_SIDE_EXTERNAL) local = true;
    if (!local) return GFSHIP_OK;
  }
  BcDesc bc;
  for (int d = 0; d < 6; d++) {
    bc.side[d] = dom->side[d];
    bc.type[d] = v->bc[d];
    bc.val[d] = (level == dom->depth) ? v->bcval[d] : nullptr;
  }
  if (!homogeneous && level != dom->depth)
    for (int d = 0; d < 2*dom->dim; d++)
      GFSHIP_CHECK (!(dom->side[d] == GFSHIP_SIDE_BOUNDARY && v->bc[d] != GFSHIP_BC_SYMMETRY &&
		      v->bcval[d]),
		    GFSHIP_EUNSUPPORTED,
		    "non-homogeneous Dirichlet/Neumann values are held on the leaf level only");
  bc.component = v->component;
  bc.homogeneous = homogeneous;
  int nface = dom->dim == 3 ? L.n*L.n : L.n;
  int block = nface >= 256 ? 256 : 64;
  dim3 grid ((nface + block - 1)/block, 2*dom->dim);
  hipLaunchKernelGGL (bc_kernel, grid, dim3 (block), 0, dom->stream, L, bc, v1->lev[level]);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

int call_exchange (gfship_domain * dom, double * ptr, int level, int kind)
{
  if (!dom->has_external)
    return GFSHIP_OK;
  if (dom->comm)
    return comm_exchange (dom, ptr, level, kind);
  GFSHIP_CHECK (dom->exchange != nullptr, GFSHIP_EINVAL,
		"the domain has GFSHIP_SIDE_EXTERNAL sides but neither a communicator "
		"(gfship_domain_comm_init) nor an exchange hook (gfship_domain_set_exchange)");
  int r = (* dom->exchange) (dom->exchange_ctx, ptr, level, kind);
  GFSHIP_CHECK (r == 0, GFSHIP_EHIP, "the exchange hook failed (%d)", r);
  return GFSHIP_OK;
}

// MPI_Allgather of a device array: `count' doubles of every box, in rank order
int call_gather (gfship_domain * dom, const double * send, double * recv, size_t count)
{
  if (dom->comm)
    return comm_allgather (dom, send, recv, count);
  GFSHIP_CHECK (dom->gather != nullptr, GFSHIP_EINVAL, "no communicator and no gather hook");
  int r = (* dom->gather) (dom->gather_ctx, send, recv, count*sizeof (double));
  GFSHIP_CHECK (r == 0, GFSHIP_EHIP, "the gather hook failed (%d)", r);
  return GFSHIP_OK;
}

int call_reduce (gfship_domain * dom, double * vals, int n, int op)
{
  if (!dom->has_external)
    return GFSHIP_OK;
  if (dom->comm)
    return op == 0 ? comm_reduce (dom, vals, n, nullptr, 0, nullptr, 0) :
      op == 1 ? comm_reduce (dom, nullptr, 0, vals, n, nullptr, 0) :
      comm_reduce (dom, nullptr, 0, nullptr, 0, vals, n);
  GFSHIP_CHECK (dom->reduce != nullptr, GFSHIP_EINVAL,
		"the domain has GFSHIP_SIDE_EXTERNAL sides but neither a communicator "
		"(gfship_domain_comm_init) nor a reduce hook (gfship_domain_set_reduce)");
  int r = (* dom->reduce) (dom->reduce_ctx, vals, n, op);
  GFSHIP_CHECK (r == 0, GFSHIP_EHIP, "the reduce hook failed (%d)", r);
  return GFSHIP_OK;
}

// domain_norm_reduce (src/domain.c:2135-2166): nsum sums and one maximum; one collective with the
// in-library transport, one hook call per operation otherwise
int call_reduce_norm (gfship_domain * dom, double * sums, int nsum, double * mx)
{
  if (!dom->has_external)
    return GFSHIP_OK;
  if (dom->comm)
    return comm_reduce (dom, sums, nsum, mx, 1, nullptr, 0);
  int r;
  if ((r = call_reduce (dom, sums, nsum, 0))) return r;
  return call_reduce (dom, mx, 1, 1);
}

// sndbuf / rcvbuf of the periodic and MPI boundaries (src/boundary.c:1240-1258,1333-1347):
// interior layer of `side` -> contiguous buffer, buffer -> ghost layer of `side`
__global__ void __launch_bounds__(256)
halo_copy_kernel (Layout L, int side, double * __restrict__ a, double * __restrict__ buf, int unpack)
{
  const int n = L.n;
  const int nface = L.dim == 3 ? n*n : n;
  int f = blockIdx.x*blockDim.x + threadIdx.x;
  if (f >= nface) return;
  int c = side/2;
  int t1 = f % n + 1, t2 = L.dim == 3 ? f / n + 1 : 0;
  int ijk[3] = { 0, 0, 0 };
  int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
  ijk[c] = (side & 1) ? 1 : n;
  ijk[ta] = t1;
  if (L.dim == 3) ijk[tb] = t2;
  long o = c == 0 ? 1 : c == 1 ? L.sy : L.sz;
  if (side & 1) o = - o;
  long nb = L.idx (ijk[0], ijk[1], ijk[2]);
  if (unpack)
    a[nb + o] = buf[f];
  else
    buf[f] = a[nb];
}

// the same for several sides in one launch: blockIdx.y = entry of the side list
struct HaloSides { int n; int side[6]; double * buf[6]; };

__global__ void __launch_bounds__(256)
halo_copy_sides_kernel (Layout L, HaloSides H, double * __restrict__ a, int unpack)
{
  const int n = L.n;
  const int nface = L.dim == 3 ? n*n : n;
  int f = blockIdx.x*blockDim.x + threadIdx.x;
  if (f >= nface) return;
  const int side = H.side[blockIdx.y];
  double * __restrict__ buf = H.buf[blockIdx.y];
  int c = side/2;
  int t1 = f % n + 1, t2 = L.dim == 3 ? f / n + 1 : 0;
  int ijk[3] = { 0, 0, 0 };
  int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
  ijk[c] = (side & 1) ? 1 : n;
  ijk[ta] = t1;
  if (L.dim == 3) ijk[tb] = t2;
  long o = c == 0 ? 1 : c == 1 ? L.sy : L.sz;
  if (side & 1) o = - o;
  long nb = L.idx (ijk[0], ijk[1], ijk[2]);
  if (unpack)
    a[nb + o] = buf[f];
  else
    buf[f] = a[nb];
}

static int halo_copy_sides (gfship_domain * dom, double * a, int level, int nsides, const int * sides,
			    void * const * bufs, int unpack)
{
  GFSHIP_CHECK (dom && a && sides && bufs, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (level >= 0 && level <= dom->depth, GFSHIP_EINVAL, "level %d out of range", level);
  GFSHIP_CHECK (nsides >= 0 && nsides <= 6, GFSHIP_EINVAL, "at most six sides");
  if (nsides == 0) return GFSHIP_OK;
  HaloSides H;
  H.n = nsides;
  for (int q = 0; q < nsides; q++) {
    GFSHIP_CHECK (sides[q] >= 0 && sides[q] < 2*dom->dim && bufs[q], GFSHIP_EINVAL,
		  "side %d out of range or null buffer", sides[q]);
    H.side[q] = sides[q];
    H.buf[q] = (double *) bufs[q];
  }
  const Layout & L = dom->lay[level];
  int nface = dom->dim == 3 ? L.n*L.n : L.n;
  int block = nface >= 256 ? 256 : 64;
  hipLaunchKernelGGL (halo_copy_sides_kernel, dim3 ((nface + block - 1)/block, nsides), dim3 (block), 0,
		      dom->stream, L, H, a, unpack);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

static int halo_copy (gfship_domain * dom, double * a, int level, int side, double * buf, int unpack)
{
  GFSHIP_CHECK (dom && a && buf, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (level >= 0 && level <= dom->depth, GFSHIP_EINVAL, "level %d out of range", level);
  GFSHIP_CHECK (side >= 0 && side < 2*dom->dim, GFSHIP_EINVAL, "side %d out of range", side);
  const Layout & L = dom->lay[level];
  int nface = dom->dim == 3 ? L.n*L.n : L.n;
  int block = nface >= 256 ? 256 : 64;
  hipLaunchKernelGGL (halo_copy_kernel, dim3 ((nface + block - 1)/block), dim3 (block), 0, dom->stream,
		      L, side, a, buf, unpack);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// The BC application of up to three variables at once (the velocity components, the components of
// a pressure gradient: gfs_domain_bc is called for them one after the other, src/timestep.c:
// 85,527): on MPI sides one pack kernel, ONE message per side carrying the layers of all of them,
// one unpack kernel.
// ---------------------------------------------------------------------------------------------
struct HaloMulti { int n; int side[6]; double * buf[6]; double * a[3]; int nf; };

// blockIdx.y = entry of the side list, blockIdx.z = variable: its layer sits at buf + z nface
__global__ void __launch_bounds__(256)
halo_copy_multi_kernel (Layout L, HaloMulti H, int unpack)
{
  const int n = L.n;
  const int nface = L.dim == 3 ? n*n : n;
  int f = blockIdx.x*blockDim.x + threadIdx.x;
  if (f >= nface) return;
  const int side = H.side[blockIdx.y];
  double * __restrict__ buf = H.buf[blockIdx.y] + (size_t) blockIdx.z*nface;
  double * __restrict__ a = H.a[blockIdx.z];
  int c = side/2;
  int t1 = f % n + 1, t2 = L.dim == 3 ? f / n + 1 : 0;
  int ijk[3] = { 0, 0, 0 };
  int ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
  ijk[c] = (side & 1) ? 1 : n;
  ijk[ta] = t1;
  if (L.dim == 3) ijk[tb] = t2;
  long o = c == 0 ? 1 : c == 1 ? L.sy : L.sz;
  if (side & 1) o = - o;
  long nb = L.idx (ijk[0], ijk[1], ijk[2]);
  if (unpack)
    a[nb + o] = buf[f];
  else
    buf[f] = a[nb];
}

// send / receive buffers of 3 n^2 doubles per MPI side (leaf level): several variables in one
// message, the face states of the tiled Godunov kernels
int multi_buffers (gfship_domain * dom)
{
  const Layout & L = dom->lay[dom->depth];
  const size_t bytes = (size_t) 3*(dom->dim == 3 ? (size_t) L.n*L.n : (size_t) L.n)*sizeof (double);
  for (int d = 0; d < 2*dom->dim; d++) {
    if (dom->side[d] != GFSHIP_SIDE_EXTERNAL) continue;
    if (!dom->gfv_send[d]) GFSHIP_HIP (hipMalloc ((void **) &dom->gfv_send[d], bytes));
    if (!dom->gfv_recv[d]) GFSHIP_HIP (hipMalloc ((void **) &dom->gfv_recv[d], bytes));
  }
  return GFSHIP_OK;
}

int launch_bc_multi (gfship_domain * dom, Field * const * v, int nf, int level, int homogeneous)
{
  GFSHIP_CHECK (nf >= 1 && nf <= 3, GFSHIP_EINVAL, "one to three variables");
  const Layout & L = dom->lay[level];
  const int nface = dom->dim == 3 ? L.n*L.n : L.n;
  const int block = nface >= 256 ? 256 : 64;
  /* the local sides: one launch per variable (measured: a three-variable kernel is no faster) */
  for (int q = 0; q < nf; q++)
    if (int r = launch_bc_kernel (dom, v[q], v[q], level, homogeneous)) return r;
  if (!dom->has_external)
    return GFSHIP_OK;
  if (!dom->comm || nf == 1) {
    for (int q = 0; q < nf; q++)
      if (int r = call_exchange (dom, v[q]->lev[level], level, 0)) return r;
    return GFSHIP_OK;
  }
  if (int r = multi_buffers (dom)) return r;
  HaloMulti H;
  H.n = 0; H.nf = nf;
  for (int q = 0; q < nf; q++) H.a[q] = v[q]->lev[level];
  for (int d = 0; d < 2*dom->dim; d++)
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL) {
      H.side[H.n] = d;
      H.buf[H.n++] = dom->gfv_send[d];
    }
  hipLaunchKernelGGL (halo_copy_multi_kernel, dim3 ((nface + block - 1)/block, H.n, nf), dim3 (block), 0,
		      dom->stream, L, H, 0);
  GFSHIP_HIP (hipGetLastError ());
  if (int r = comm_exchange_raw (dom, dom->gfv_send, dom->gfv_recv, (size_t) nf*nface)) return r;
  /* the layer received across side d fills the ghost layer of side d */
  for (int q = 0; q < H.n; q++)
    H.buf[q] = dom->gfv_recv[H.side[q]];
  hipLaunchKernelGGL (halo_copy_multi_kernel, dim3 ((nface + block - 1)/block, H.n, nf), dim3 (block), 0,
		      dom->stream, L, H, 1);
  GFSHIP_HIP (hipGetLastError ());
  return GFSHIP_OK;
}

} // namespace gfship

extern "C" {

int gfship_halo_pack (gfship_domain * dom, const void * dev_ptr, int level, int side, void * dev_buf)
{
  return gfship::halo_copy (dom, (double *) dev_ptr, level, side, (double *) dev_buf, 0);
}

int gfship_halo_unpack (gfship_domain * dom, void * dev_ptr, int level, int side, const void * dev_buf)
{
  return gfship::halo_copy (dom, (double *) dev_ptr, level, side, (double *) dev_buf, 1);
}

int gfship_halo_pack_sides (gfship_domain * dom, const void * dev_ptr, int level, int nsides,
			    const int * sides, void * const * dev_bufs)
{
  return gfship::halo_copy_sides (dom, (double *) dev_ptr, level, nsides, sides, dev_bufs, 0);
}

int gfship_halo_unpack_sides (gfship_domain * dom, void * dev_ptr, int level, int nsides,
			      const int * sides, void * const * dev_bufs)
{
  return gfship::halo_copy_sides (dom, (double *) dev_ptr, level, nsides, sides, dev_bufs, 1);
}

} // extern "C"

namespace gfship {


} // namespace gfship
