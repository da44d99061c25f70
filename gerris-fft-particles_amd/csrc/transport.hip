// transport.hip -- GfsBoundaryMpi over RCCL, inside the library: the halo exchange of every BC
// application and the all-reduces of norms / CFL of a domain that is one GfsBox of a periodic
// lattice of boxes, one box per GPU (src/mpi_boundary.c:78-246, src/domain.c:2135-2166,2921).
//
// One exchange = pack kernel (all sides, one launch) -> one ncclGroupStart/End with an ncclSend per
// MPI side and the matching ncclRecv -> unpack kernel, all enqueued on the domain's stream: no
// host synchronisation, no Python, no callback.  Reductions are one ncclAllGather of the operands
// of every rank followed by the same rank-ordered reduction on every rank (identical results
// everywhere, so that all boxes take the same branches of the solve loop), read back through
// pinned host memory.
//
// RCCL is opened at run time (dlopen librccl.so.1): a single-box run never loads it, and a process
// that already holds an RCCL (PyTorch's) shares that copy.
#include "gfship_internal.hpp"
#include <cstdlib>
#include <vector>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace gfship {

struct RcclApi {
  void * handle = nullptr;
  ncclResult_t (* GetUniqueId) (ncclUniqueId *) = nullptr;
  ncclResult_t (* CommInitRank) (ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (* CommDestroy) (ncclComm_t) = nullptr;
  ncclResult_t (* CommCount) (const ncclComm_t, int *) = nullptr;
  ncclResult_t (* GroupStart) () = nullptr;
  ncclResult_t (* GroupEnd) () = nullptr;
  ncclResult_t (* Send) (const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (* Recv) (void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (* AllGather) (const void *, void *, size_t, ncclDataType_t, ncclComm_t,
			      hipStream_t) = nullptr;
  const char * (* GetErrorString) (ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int rccl_load ()
{
  if (g_rccl.handle) return GFSHIP_OK;
  /* GFSHIP_RCCL_LIBRARY: the RCCL to open (a site's own build; the test suite's in-process stand-in,
     tests/mock_rccl, which runs N ranks of this transport as N threads on one GPU): no fallback to the
     system's library when it is set and cannot be opened */
  const char * named = getenv ("GFSHIP_RCCL_LIBRARY");
  const char * names[] = { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
  void * h = nullptr;
  if (named && *named)
    h = dlopen (named, RTLD_NOW | RTLD_LOCAL);
  else
    for (const char * nm : names)
      if ((h = dlopen (nm, RTLD_NOW | RTLD_GLOBAL)))
	break;
  GFSHIP_CHECK (h != nullptr, GFSHIP_EUNSUPPORTED, "cannot open %s: %s",
		named && *named ? named : "librccl.so.1", dlerror ());
#define SYM(field, name) do { \
    *(void **) &g_rccl.field = dlsym (h, name); \
    GFSHIP_CHECK (g_rccl.field != nullptr, GFSHIP_EUNSUPPORTED, "librccl: no symbol %s", name); \
  } while (0)
  SYM (GetUniqueId, "ncclGetUniqueId");
  SYM (CommInitRank, "ncclCommInitRank");
  SYM (CommDestroy, "ncclCommDestroy");
  SYM (CommCount, "ncclCommCount");
  SYM (GroupStart, "ncclGroupStart");
  SYM (GroupEnd, "ncclGroupEnd");
  SYM (Send, "ncclSend");
  SYM (Recv, "ncclRecv");
  SYM (AllGather, "ncclAllGather");
  SYM (GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.handle = h;
  return GFSHIP_OK;
}

#define GFSHIP_NCCL(call) do { ncclResult_t e_ = (call); \
    if (e_ != ncclSuccess) { \
      gfship::set_error ("RCCL error %d (%s) in %s at %s:%d", (int) e_, \
			 g_rccl.GetErrorString (e_), #call, __FILE__, __LINE__); \
      return GFSHIP_EHIP; } } while (0)

#define COMM_MAXRED 8     /* doubles per rank in one reduction */

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  int b[3] = { 1, 1, 1 };           // lattice of boxes; rank r sits at (r % bx, (r/bx) % by, r/(bx by))
  int peer[6] = { 0, 0, 0, 0, 0, 0 }; // rank of the box across side d
  double * sbuf[6] = {}, * rbuf[6] = {}; // sndbuf / rcvbuf of the leaf level (coarser levels use a part)
  double * dred = nullptr;          // [COMM_MAXRED] operands, then [nranks*COMM_MAXRED] gathered
  double * hred = nullptr;          // pinned host mirror of dred
  unsigned long long messages = 0, bytes = 0;   // domain->mpi_messages / mpi_size, mpi_boundary.c:113,128
  // overlap = 1: the exchange of a sweep runs on a stream of its own beside the bulk of the sweep
  hipStream_t side = nullptr;
  hipEvent_t packed = nullptr, arrived = nullptr;
  // particle migration: counts (6 out, 6 in) and packets
  double * mcount = nullptr, * mbuf = nullptr;
  size_t mcap = 0;
};

static int comm_rank_of (const Comm * C, int cx, int cy, int cz)
{
  const int bx = C->b[0], by = C->b[1], bz = C->b[2];
  cx = (cx % bx + bx) % bx; cy = (cy % by + by) % by; cz = (cz % bz + bz) % bz;
  return cx + bx*(cy + by*cz);
}

// gfs_domain_copy_bc on GfsBoundaryMpi sides: kind 0 = every MPI side of a cell-centred variable;
// kind 1 + e = the face values fv[e] of gfs_domain_face_bc: my layer along side e goes out, the
// ghost layer of side e^1 comes in
int comm_exchange (gfship_domain * dom, double * a, int level, int kind)
{
  Comm * C = (Comm *) dom->comm;
  const Layout & L = dom->lay[level];
  const size_t nface = dom->dim == 3 ? (size_t) L.n*L.n : (size_t) L.n;
  int send[6], recv[6], ns = 0, nr = 0;
  if (kind == 0) {
    for (int d = 0; d < 2*dom->dim; d++)
      if (dom->side[d] == GFSHIP_SIDE_EXTERNAL) send[ns++] = d;
    /* receives in increasing order of the SENDER's side index (r^1): with two boxes along an
       axis both messages of a pair of ranks travel between the same two peers, and RCCL matches
       the sends and receives of a pair in posting order (the reference tags them with the side,
       src/mpi_boundary.c:78-83) */
    for (int q = 0; q < ns; q++) recv[nr++] = send[q] ^ 1;
  }
  else {
    const int e = kind - 1;
    if (e >= 0 && e < 2*dom->dim && dom->side[e] == GFSHIP_SIDE_EXTERNAL) {
      send[ns++] = e;
      recv[nr++] = e ^ 1;
    }
  }
  if (ns == 0) return GFSHIP_OK;
  void * sb[6], * rb[6];
  for (int q = 0; q < ns; q++) sb[q] = C->sbuf[send[q]];
  for (int q = 0; q < nr; q++) rb[q] = C->rbuf[recv[q]];
  int r = gfship_halo_pack_sides (dom, a, level, ns, send, sb);
  if (r) return r;
  GFSHIP_NCCL (g_rccl.GroupStart ());
  for (int q = 0; q < ns; q++)
    GFSHIP_NCCL (g_rccl.Send (sb[q], nface, ncclDouble, C->peer[send[q]], C->comm, dom->stream));
  for (int q = 0; q < nr; q++)
    GFSHIP_NCCL (g_rccl.Recv (rb[q], nface, ncclDouble, C->peer[recv[q]], C->comm, dom->stream));
  GFSHIP_NCCL (g_rccl.GroupEnd ());
  C->messages += ns;
  C->bytes += ns*nface*sizeof (double);
  return gfship_halo_unpack_sides (dom, a, level, nr, recv, rb);
}

// The same exchange (kind 0) in two halves, for the sweeps of a parallel relax loop with the domain
// parameter overlap = 1 (gfs_traverse_and_homogeneous_bc, src/domain.c:1109-1123): `begin' after the
// cells along the MPI sides have been swept -- their layers are packed on the domain's stream, sent
// and received on a stream of the communicator's own, beside the bulk of the sweep that goes on on
// the domain's stream (gfs_boundary_send inside update_mpi_boundaries :1024-1049) -- and `end' after
// the bulk: the domain's stream waits for the arrivals and unpacks them into the ghost layer
// (box_receive_bc / box_synchronize :1119-1122).
static int comm_sides (gfship_domain * dom, int send[6], int recv[6])
{
  int ns = 0;
  for (int d = 0; d < 2*dom->dim; d++)
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL) send[ns++] = d;
  for (int q = 0; q < ns; q++) recv[q] = send[q] ^ 1;
  return ns;
}

int comm_exchange_begin (gfship_domain * dom, double * a, int level)
{
  Comm * C = (Comm *) dom->comm;
  const Layout & L = dom->lay[level];
  const size_t nface = dom->dim == 3 ? (size_t) L.n*L.n : (size_t) L.n;
  int send[6], recv[6];
  const int ns = comm_sides (dom, send, recv);
  if (ns == 0) return GFSHIP_OK;
  void * sb[6];
  for (int q = 0; q < ns; q++) sb[q] = C->sbuf[send[q]];
  int r = gfship_halo_pack_sides (dom, a, level, ns, send, sb);
  if (r) return r;
  GFSHIP_HIP (hipEventRecord (C->packed, dom->stream));
  GFSHIP_HIP (hipStreamWaitEvent (C->side, C->packed, 0));
  GFSHIP_NCCL (g_rccl.GroupStart ());
  for (int q = 0; q < ns; q++)
    GFSHIP_NCCL (g_rccl.Send (sb[q], nface, ncclDouble, C->peer[send[q]], C->comm, C->side));
  for (int q = 0; q < ns; q++)
    GFSHIP_NCCL (g_rccl.Recv (C->rbuf[recv[q]], nface, ncclDouble, C->peer[recv[q]], C->comm, C->side));
  GFSHIP_NCCL (g_rccl.GroupEnd ());
  GFSHIP_HIP (hipEventRecord (C->arrived, C->side));
  C->messages += ns;
  C->bytes += ns*nface*sizeof (double);
  return GFSHIP_OK;
}

int comm_exchange_end (gfship_domain * dom, double * a, int level)
{
  Comm * C = (Comm *) dom->comm;
  int send[6], recv[6];
  const int ns = comm_sides (dom, send, recv);
  if (ns == 0) return GFSHIP_OK;
  void * rb[6];
  for (int q = 0; q < ns; q++) rb[q] = C->rbuf[recv[q]];
  GFSHIP_HIP (hipStreamWaitEvent (dom->stream, C->arrived, 0));
  return gfship_halo_unpack_sides (dom, a, level, ns, recv, rb);
}

// MPI_Allreduce of nsum sums, nmax maxima and nmin minima in one collective: every rank gathers
// the operands of all ranks and reduces them in rank order
int comm_reduce (gfship_domain * dom, double * sums, int nsum, double * maxs, int nmax,
		 double * mins, int nmin)
{
  Comm * C = (Comm *) dom->comm;
  const int n = nsum + nmax + nmin;
  GFSHIP_CHECK (n > 0 && n <= COMM_MAXRED, GFSHIP_EINVAL, "at most %d operands per reduction",
		COMM_MAXRED);
  double * h = C->hred;
  for (int q = 0; q < nsum; q++) h[q] = sums[q];
  for (int q = 0; q < nmax; q++) h[nsum + q] = maxs[q];
  for (int q = 0; q < nmin; q++) h[nsum + nmax + q] = mins[q];
  GFSHIP_HIP (hipMemcpyAsync (C->dred, h, n*sizeof (double), hipMemcpyHostToDevice, dom->stream));
  GFSHIP_NCCL (g_rccl.AllGather (C->dred, C->dred + COMM_MAXRED, (size_t) n, ncclDouble, C->comm,
				 dom->stream));
  GFSHIP_HIP (hipMemcpyAsync (h + COMM_MAXRED, C->dred + COMM_MAXRED,
			      (size_t) C->nranks*n*sizeof (double), hipMemcpyDeviceToHost, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  const double * g = h + COMM_MAXRED;
  for (int q = 0; q < n; q++) {
    double v = g[q];
    for (int rk = 1; rk < C->nranks; rk++) {
      const double x = g[(size_t) rk*n + q];
      if (q < nsum) v += x;
      else if (q < nsum + nmax) v = x > v ? x : v;
      else v = x < v ? x : v;
    }
    if (q < nsum) sums[q] = v;
    else if (q < nsum + nmax) maxs[q - nsum] = v;
    else mins[q - nsum - nmax] = v;
  }
  return GFSHIP_OK;
}

// `count' doubles per MPI side from caller's buffers: send[d] goes to the box across side d, recv[d]
// receives what that box sent across its side d ^ 1 (same matching order as comm_exchange)
int comm_exchange_raw (gfship_domain * dom, double * const send[6], double * const recv[6], size_t count)
{
  Comm * C = (Comm *) dom->comm;
  int sd[6], rv[6];
  const int ns = comm_sides (dom, sd, rv);
  if (ns == 0) return GFSHIP_OK;
  GFSHIP_NCCL (g_rccl.GroupStart ());
  for (int q = 0; q < ns; q++)
    GFSHIP_NCCL (g_rccl.Send (send[sd[q]], count, ncclDouble, C->peer[sd[q]], C->comm, dom->stream));
  for (int q = 0; q < ns; q++)
    GFSHIP_NCCL (g_rccl.Recv (recv[rv[q]], count, ncclDouble, C->peer[rv[q]], C->comm, dom->stream));
  GFSHIP_NCCL (g_rccl.GroupEnd ());
  C->messages += ns;
  C->bytes += ns*count*sizeof (double);
  return GFSHIP_OK;
}

// send_particles / rcv_particles (modules/particulatecommon.c:3218-3312) over the communicator: the
// packets of particles that leave through each MPI side go to the box across it, the packets of
// the neighbours come in -- counts first, then the payloads, each one ncclSend / ncclRecv pair per
// side in one group, staged through device buffers (the lists are rebuilt on the host side of the
// particle code, which sorts the packets by id)
int comm_migrate (gfship_domain * dom, int rs, const int nsend[6], const double * const send[6],
		  int nrecv[6], std::vector<double> recv[6])
{
  Comm * C = (Comm *) dom->comm;
  int sd[6], rv[6];
  const int ns = comm_sides (dom, sd, rv);
  for (int d = 0; d < 6; d++) { nrecv[d] = 0; recv[d].clear (); }
  if (ns == 0) return GFSHIP_OK;
  // counts: one double per side
  double hc[12];
  for (int d = 0; d < 6; d++) { hc[d] = (double) nsend[d]; hc[6 + d] = 0.; }
  if (!C->mcount)
    GFSHIP_HIP (hipMalloc ((void **) &C->mcount, 12*sizeof (double)));
  GFSHIP_HIP (hipMemcpyAsync (C->mcount, hc, 12*sizeof (double), hipMemcpyHostToDevice, dom->stream));
  GFSHIP_NCCL (g_rccl.GroupStart ());
  for (int q = 0; q < ns; q++)
    GFSHIP_NCCL (g_rccl.Send (C->mcount + sd[q], 1, ncclDouble, C->peer[sd[q]], C->comm, dom->stream));
  for (int q = 0; q < ns; q++)
    GFSHIP_NCCL (g_rccl.Recv (C->mcount + 6 + rv[q], 1, ncclDouble, C->peer[rv[q]], C->comm, dom->stream));
  GFSHIP_NCCL (g_rccl.GroupEnd ());
  GFSHIP_HIP (hipMemcpyAsync (hc, C->mcount, 12*sizeof (double), hipMemcpyDeviceToHost, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  size_t tot_s = 0, tot_r = 0, os[6], orr[6];
  for (int d = 0; d < 6; d++) {
    nrecv[d] = (int) hc[6 + d];
    os[d] = tot_s; tot_s += (size_t) nsend[d]*rs;
    orr[d] = tot_r; tot_r += (size_t) nrecv[d]*rs;
  }
  C->messages += 2*ns;
  C->bytes += (tot_s + ns)*sizeof (double);
  if (tot_s + tot_r == 0) return GFSHIP_OK;
  if (C->mcap < tot_s + tot_r) {
    if (C->mbuf) GFSHIP_HIP (hipFree (C->mbuf));
    C->mbuf = nullptr; C->mcap = 0;
    GFSHIP_HIP (hipMalloc ((void **) &C->mbuf, 2*(tot_s + tot_r)*sizeof (double)));
    C->mcap = 2*(tot_s + tot_r);
  }
  double * ds = C->mbuf, * dr = C->mbuf + tot_s;
  for (int d = 0; d < 6; d++)
    if (nsend[d] > 0)
      GFSHIP_HIP (hipMemcpyAsync (ds + os[d], send[d], (size_t) nsend[d]*rs*sizeof (double),
				  hipMemcpyHostToDevice, dom->stream));
  GFSHIP_NCCL (g_rccl.GroupStart ());
  for (int q = 0; q < ns; q++)
    if (nsend[sd[q]] > 0)
      GFSHIP_NCCL (g_rccl.Send (ds + os[sd[q]], (size_t) nsend[sd[q]]*rs, ncclDouble, C->peer[sd[q]], C->comm,
				dom->stream));
  for (int q = 0; q < ns; q++)
    if (nrecv[rv[q]] > 0)
      GFSHIP_NCCL (g_rccl.Recv (dr + orr[rv[q]], (size_t) nrecv[rv[q]]*rs, ncclDouble, C->peer[rv[q]], C->comm,
				dom->stream));
  GFSHIP_NCCL (g_rccl.GroupEnd ());
  for (int d = 0; d < 6; d++)
    if (nrecv[d] > 0) {
      recv[d].resize ((size_t) nrecv[d]*rs);
      GFSHIP_HIP (hipMemcpyAsync (recv[d].data (), dr + orr[d], (size_t) nrecv[d]*rs*sizeof (double),
				  hipMemcpyDeviceToHost, dom->stream));
    }
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  return GFSHIP_OK;
}

// MPI_Allgather of `count' doubles per rank, in rank order, on the domain's stream
int comm_allgather (gfship_domain * dom, const double * send, double * recv, size_t count)
{
  Comm * C = (Comm *) dom->comm;
  GFSHIP_NCCL (g_rccl.AllGather (send, recv, count, ncclDouble, C->comm, dom->stream));
  C->messages += 1;
  C->bytes += count*sizeof (double);
  return GFSHIP_OK;
}

void comm_free (gfship_domain * dom)
{
  Comm * C = (Comm *) dom->comm;
  if (!C) return;
  if (dom->stream) (void) hipStreamSynchronize (dom->stream);
  if (C->side) (void) hipStreamSynchronize (C->side);
  if (C->comm && g_rccl.CommDestroy) (void) g_rccl.CommDestroy (C->comm);
  if (C->packed) (void) hipEventDestroy (C->packed);
  if (C->arrived) (void) hipEventDestroy (C->arrived);
  if (C->side) (void) hipStreamDestroy (C->side);
  for (int d = 0; d < 6; d++) {
    if (C->sbuf[d]) (void) hipFree (C->sbuf[d]);
    if (C->rbuf[d]) (void) hipFree (C->rbuf[d]);
  }
  if (C->dred) (void) hipFree (C->dred);
  if (C->mcount) (void) hipFree (C->mcount);
  if (C->mbuf) (void) hipFree (C->mbuf);
  if (C->hred) (void) hipHostFree (C->hred);
  delete C;
  dom->comm = nullptr;
}

} // namespace gfship

using namespace gfship;

extern "C" {

int gfship_comm_available (void)
{
  return rccl_load ();
}

int gfship_comm_unique_id (void * id)
{
  GFSHIP_CHECK (id != nullptr, GFSHIP_EINVAL, "null pointer");
  int r = rccl_load ();
  if (r) return r;
  ncclUniqueId u;
  GFSHIP_NCCL (g_rccl.GetUniqueId (&u));
  memcpy (id, &u, GFSHIP_UNIQUE_ID_BYTES);
  return GFSHIP_OK;
}

int gfship_domain_comm_init (gfship_domain * dom, const void * unique_id, int rank, int nranks,
			     const int lattice[3])
{
  GFSHIP_CHECK (dom && unique_id && lattice, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (dom->comm == nullptr, GFSHIP_EINVAL, "the domain already has a communicator");
  GFSHIP_CHECK (nranks >= 1 && rank >= 0 && rank < nranks, GFSHIP_EINVAL, "rank %d of %d", rank,
		nranks);
  long nb = 1;
  for (int c = 0; c < 3; c++) {
    GFSHIP_CHECK (lattice[c] >= 1 && (c < dom->dim || lattice[c] == 1), GFSHIP_EINVAL,
		  "lattice[%d] = %d", c, lattice[c]);
    nb *= lattice[c];
  }
  GFSHIP_CHECK (nb == nranks, GFSHIP_EINVAL, "a lattice of %d x %d x %d boxes needs %ld ranks, not %d",
		lattice[0], lattice[1], lattice[2], nb, nranks);
  for (int d = 0; d < 2*dom->dim; d++)
    GFSHIP_CHECK (lattice[d/2] == 1 || dom->side[d] == GFSHIP_SIDE_EXTERNAL, GFSHIP_EINVAL,
		  "side %d faces another box of the lattice: it must be GFSHIP_SIDE_EXTERNAL", d);
  int r = rccl_load ();
  if (r) return r;
  GFSHIP_HIP (hipSetDevice (dom->device));
  Comm * C = new Comm;
  dom->comm = C;
  C->rank = rank; C->nranks = nranks;
  for (int c = 0; c < 3; c++) C->b[c] = lattice[c];
  dom->lat_rank = rank; dom->lat_n = nranks;
  for (int c = 0; c < 3; c++) dom->lat_b[c] = lattice[c];
  const int cx = rank % C->b[0], cy = (rank/C->b[0]) % C->b[1], cz = rank/(C->b[0]*C->b[1]);
  for (int d = 0; d < 6; d++) {
    int cc[3] = { cx, cy, cz };
    cc[d/2] += (d & 1) ? -1 : 1;
    C->peer[d] = comm_rank_of (C, cc[0], cc[1], cc[2]);
  }
  ncclUniqueId u;
  memcpy (&u, unique_id, GFSHIP_UNIQUE_ID_BYTES);
  ncclResult_t e = g_rccl.CommInitRank (&C->comm, nranks, u, rank);
  if (e != ncclSuccess) {
    set_error ("ncclCommInitRank failed: %d (%s)", (int) e, g_rccl.GetErrorString (e));
    C->comm = nullptr;
    comm_free (dom);
    return GFSHIP_EHIP;
  }
  const Layout & L = dom->lay[dom->depth];
  const size_t nface = dom->dim == 3 ? (size_t) L.n*L.n : (size_t) L.n;
  hipError_t he = hipSuccess;
  for (int d = 0; d < 2*dom->dim && he == hipSuccess; d++)
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL) {
      he = hipMalloc ((void **) &C->sbuf[d], nface*sizeof (double));
      if (he == hipSuccess) he = hipMalloc ((void **) &C->rbuf[d], nface*sizeof (double));
    }
  if (he == hipSuccess) he = hipStreamCreateWithFlags (&C->side, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipEventCreateWithFlags (&C->packed, hipEventDisableTiming);
  if (he == hipSuccess) he = hipEventCreateWithFlags (&C->arrived, hipEventDisableTiming);
  if (he == hipSuccess)
    he = hipMalloc ((void **) &C->dred, (size_t) (nranks + 1)*COMM_MAXRED*sizeof (double));
  if (he == hipSuccess)
    he = hipHostMalloc ((void **) &C->hred, (size_t) (nranks + 1)*COMM_MAXRED*sizeof (double),
			hipHostMallocDefault);
  if (he != hipSuccess) {
    int rr = hip_fail (he, "communicator buffers", __FILE__, __LINE__);
    comm_free (dom);
    return rr;
  }
  return GFSHIP_OK;
}

int gfship_domain_comm_size (gfship_domain * dom)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  if (!dom->comm) return 0;
  int n = 0;
  GFSHIP_NCCL (g_rccl.CommCount (((Comm *) dom->comm)->comm, &n));
  return n;
}

int gfship_domain_comm_stats (gfship_domain * dom, unsigned long long * messages,
			      unsigned long long * bytes)
{
  GFSHIP_CHECK (dom && dom->comm, GFSHIP_EINVAL, "the domain has no communicator");
  if (messages) *messages = ((Comm *) dom->comm)->messages;
  if (bytes) *bytes = ((Comm *) dom->comm)->bytes;
  return GFSHIP_OK;
}

int gfship_domain_comm_destroy (gfship_domain * dom)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  comm_free (dom);
  return GFSHIP_OK;
}

} // extern "C"
