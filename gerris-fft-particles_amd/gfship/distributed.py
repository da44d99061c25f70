"""One GfsBox per rank: box layout, neighbour topology and the halo exchange / all-reduce hooks
over torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPUs).

Mirrors the reference's parallel model (src/mpi_boundary.c:78-246, src/domain.c:2135-2166):
  * boxes tile a periodic bx x by x bz lattice (gfs_domain_split / one box per PE); a side whose
    lattice extent is 1 stays a local periodic side, the others become GfsBoundaryMpi sides;
  * every BC application sends the interior layer along each MPI side to the neighbour across
    that side and receives the neighbour's layer into the ghost layer (send :89-130,
    receive :132-222); per peer the messages are matched in the order of the sender's side index
    (the reference uses a tag built from the side, :78-83);
  * norms and the CFL time step are all-reduced (sum / max / min).

bench.py uses the library's own RCCL transport (gfship_domain_comm_init, csrc/transport.hip); the
hook classes here serve the tests (tests/multibox.py drives the CPU oracle and several device boxes
of one process with the same BoxGrid / Transport) and the host-staged rehearsal of several ranks on
one GPU.
"""
import ctypes as C

import numpy as np

SIDE_PERIODIC, SIDE_BOUNDARY, SIDE_EXTERNAL = 0, 1, 2


def lattice(nboxes, dim=3):
    """bx, by, bz for 1, 2, 4, 8 ... boxes: doubled along x, then y, then z."""
    b = [1, 1, 1]
    a = 0
    n = nboxes
    while n > 1:
        if n % 2:
            raise ValueError("the number of boxes must be a power of two (got %d)" % nboxes)
        b[a % dim] *= 2
        a += 1
        n //= 2
    return tuple(b)


class BoxGrid:
    """Periodic lattice of boxes; rank r sits at (r % bx, (r // bx) % by, r // (bx*by))."""

    def __init__(self, nboxes, dim=3):
        self.dim = dim
        self.n = nboxes
        self.b = lattice(nboxes, dim)

    def coords(self, rank):
        bx, by, bz = self.b
        return (rank % bx, (rank // bx) % by, rank // (bx * by))

    def rank_of(self, c):
        bx, by, bz = self.b
        return (c[0] % bx) + bx * ((c[1] % by) + by * (c[2] % bz))

    def neighbour(self, rank, side):
        c = list(self.coords(rank))
        c[side // 2] += -1 if side & 1 else 1
        return self.rank_of(c)

    def sides(self, rank=0):
        """side kinds of a box: local periodic where the lattice has one box, MPI elsewhere"""
        s = [SIDE_PERIODIC] * 6
        for d in range(2 * self.dim):
            if self.b[d // 2] > 1:
                s[d] = SIDE_EXTERNAL
        return s

    def external_sides(self):
        return [d for d in range(2 * self.dim) if self.b[d // 2] > 1]


class Transport:
    """torch.distributed point-to-point + all-reduce for one rank of a BoxGrid."""

    def __init__(self, grid, rank, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.grid, self.rank = grid, rank
        self.device = device if device is not None else torch.device("cpu")
        self._bufs = {}
        self._ops = {}
        # rehearsal of the device path without RCCL (several ranks on one GPU, which RCCL
        # refuses): device buffers are staged through the host and sent over gloo
        self.staged = self.device.type == "cuda" and dist.get_backend() == "gloo"

    def buffers(self, key, nface):
        """(send, recv) buffers per side for messages of nface doubles"""
        b = self._bufs.get((key, nface))
        if b is None:
            t = self.torch
            b = ({s: t.empty(nface, dtype=t.float64, device=self.device)
                  for s in self.grid.external_sides()},
                 {s: t.empty(nface, dtype=t.float64, device=self.device)
                  for s in self.grid.external_sides()})
            self._bufs[(key, nface)] = b
        return b

    def exchange(self, send_sides, snd, recv_sides, rcv):
        """send snd[s] (interior layer along my side s) to the neighbour across s, for s in
        send_sides; receive into rcv[r], for r in recv_sides, the layer that the neighbour across
        my side r sent from its side r^1."""
        dist = self.dist
        if self.staged:
            hs = {s: snd[s].cpu() for s in send_sides}
            hr = {r: self.torch.empty(rcv[r].shape, dtype=rcv[r].dtype) for r in recv_sides}
            ops = [dist.P2POp(dist.isend, hs[s], self.grid.neighbour(self.rank, s), tag=s)
                   for s in sorted(send_sides)]
            ops += [dist.P2POp(dist.irecv, hr[r], self.grid.neighbour(self.rank, r), tag=r ^ 1)
                    for r in sorted(recv_sides, key=lambda x: x ^ 1)]
            for q in dist.batch_isend_irecv(ops):
                q.wait()
            for r in recv_sides:
                rcv[r].copy_(hr[r])
            return
        # the descriptors only name persistent buffers: built once per (buffers, sides)
        key = (id(snd), id(rcv), tuple(send_sides), tuple(recv_sides))
        ops = self._ops.get(key)
        if ops is None:
            ops = []
            # sends in increasing side index; receives in increasing order of the SENDER's side
            # index (r^1), so that the messages between two ranks match pairwise in posting order
            for s in sorted(send_sides):
                ops.append(dist.P2POp(dist.isend, snd[s], self.grid.neighbour(self.rank, s), tag=s))
            for r in sorted(recv_sides, key=lambda x: x ^ 1):
                ops.append(dist.P2POp(dist.irecv, rcv[r], self.grid.neighbour(self.rank, r), tag=r ^ 1))
            self._ops[key] = ops
        for q in dist.batch_isend_irecv(ops):
            q.wait()

    def exchange_records(self, out, rs=7):
        """out[s]: (n, rs) records leaving through my MPI side s; returns {r: records arriving
        through my side r}.  Counts first, then the payloads, matched like the halo messages."""
        t, dist = self.torch, self.dist
        sides = sorted(out)
        dev = "cpu" if self.staged else self.device
        cnt_s = {s: t.tensor([len(out[s])], dtype=t.int64, device=dev) for s in sides}
        cnt_r = {s: t.zeros(1, dtype=t.int64, device=dev) for s in sides}
        if self.staged:
            ops = [dist.P2POp(dist.isend, cnt_s[s], self.grid.neighbour(self.rank, s), tag=50 + s)
                   for s in sides]
            ops += [dist.P2POp(dist.irecv, cnt_r[r], self.grid.neighbour(self.rank, r), tag=50 + (r ^ 1))
                    for r in sorted(sides, key=lambda x: x ^ 1)]
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        else:
            self.exchange(sides, cnt_s, sides, cnt_r)
        snd = {s: t.from_numpy(np.ascontiguousarray(out[s]).ravel()).to(dev)
               for s in sides if len(out[s])}
        rcv = {r: t.empty(rs * int(cnt_r[r].item()), dtype=t.float64, device=dev)
               for r in sides if int(cnt_r[r].item())}
        # every box knows which of its messages are empty on both ends: only the others are posted
        ops = []
        for s in sorted(snd):
            ops.append(dist.P2POp(dist.isend, snd[s], self.grid.neighbour(self.rank, s), tag=100 + s))
        for r in sorted(rcv, key=lambda x: x ^ 1):
            ops.append(dist.P2POp(dist.irecv, rcv[r], self.grid.neighbour(self.rank, r), tag=100 + (r ^ 1)))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return {r: rcv[r].cpu().numpy().reshape(-1, rs) for r in rcv}

    def allreduce(self, vals, op):
        t, dist = self.torch, self.dist
        x = t.tensor(vals, dtype=t.float64, device="cpu" if self.staged else self.device)
        dist.all_reduce(x, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[op])
        return x.cpu().numpy()


def _kind_sides(grid, kind):
    """kind 0: every MPI side; kind 1 + e (face values fv[e]): my side e is sent, side e^1 filled"""
    ext = grid.external_sides()
    if kind == 0:
        return ext
    e = kind - 1
    return [e] if e in ext else []


# ---------------------------------------------------------------------------------------------
# hooks for libgfship (device)
# ---------------------------------------------------------------------------------------------

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class DeviceHooks:
    """Installs the exchange / reduce hooks of a gfship.Domain (include/gfship.h)."""

    def __init__(self, dom, transport):
        import gfship
        self.dom, self.tr = dom, transport
        self.lib = gfship.lib()
        torch = transport.torch
        # run torch's work on the library's stream: pack kernel -> send/recv -> unpack kernel are
        # then ordered on one stream without host synchronisation
        self.stream = torch.cuda.ExternalStream(int(self.lib.gfship_domain_stream(dom.ptr)),
                                                device=transport.device)
        self._plans = {}
        self._ex = EXCHANGE_FN(self._exchange)
        self._red = REDUCE_FN(self._reduce)
        gfship._check(self.lib.gfship_domain_set_exchange(dom.ptr, C.cast(self._ex, C.c_void_p), None))
        gfship._check(self.lib.gfship_domain_set_reduce(dom.ptr, C.cast(self._red, C.c_void_p), None))
        # a transport that can gather device memory of all boxes (gather_device (send, recv, nbytes,
        # stream)) also serves the replicated coarse end of the V-cycle (include/gfship.h)
        self._gat = None
        if hasattr(transport, "gather_device"):
            self._gat = GATHER_FN(self._gather)
            g = transport.grid
            gfship._check(self.lib.gfship_domain_set_gather(
                dom.ptr, C.cast(self._gat, C.c_void_p), None, transport.rank, g.n, (C.c_int * 3)(*g.b)))

    def _plan(self, level, kind):
        """what one exchange of (level, kind) needs, built once: side lists, buffers and the
        argument arrays of the pack / unpack calls"""
        key = (level, kind)
        p = self._plans.get(key)
        if p is None:
            sides = _kind_sides(self.tr.grid, kind)       # sides I send from
            if not sides:
                p = False
            else:
                nface = (1 << level) ** (self.dom.dim - 1)
                snd, rcv = self.tr.buffers("dev", nface)
                # kind 0: receive on the same set of sides; kind 1+e: receive on side e^1 only
                recv_sides = [x ^ 1 for x in sides] if kind else sides
                arr = lambda xs: (C.c_int * len(xs))(*xs)
                ptrs = lambda bufs, xs: (C.c_void_p * len(xs))(*[bufs[x].data_ptr() for x in xs])
                p = (sides, recv_sides, snd, rcv, arr(sides), ptrs(snd, sides), arr(recv_sides),
                     ptrs(rcv, recv_sides))
            self._plans[key] = p
        return p

    def _exchange(self, ctx, dev_ptr, level, kind):
        try:
            p = self._plan(level, kind)
            if not p:
                return 0
            sides, recv_sides, snd, rcv, s_arr, s_ptr, r_arr, r_ptr = p
            torch = self.tr.torch
            with torch.cuda.stream(self.stream):
                rc = self.lib.gfship_halo_pack_sides(self.dom.ptr, dev_ptr, level, len(sides), s_arr, s_ptr)
                if rc:
                    return rc
                self.tr.exchange(sides, snd, recv_sides, rcv)
                rc = self.lib.gfship_halo_unpack_sides(self.dom.ptr, dev_ptr, level, len(recv_sides),
                                                       r_arr, r_ptr)
                if rc:
                    return rc
            return 0
        except Exception as e:     # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    def _gather(self, ctx, send, recv, nbytes):
        try:
            self.tr.gather_device(send, recv, nbytes, int(self.lib.gfship_domain_stream(self.dom.ptr)))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _reduce(self, ctx, vals, n, op):
        try:
            a = np.ctypeslib.as_array(vals, shape=(n,))
            with self.tr.torch.cuda.stream(self.stream):
                a[...] = self.tr.allreduce(a.copy(), op)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1


MIGRATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)),
                         C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)))


class ParticleMigration:
    """Installs the migration hook of a gfship.ParticleList: the packets of particles that left
    through the MPI sides go to the neighbour boxes, theirs come in
    (modules/particulatecommon.c:3218-3312)."""

    def __init__(self, plist, transport):
        import gfship
        self.pl, self.tr = plist, transport
        self._fn = MIGRATE_FN(self._migrate)
        self._keep = {}
        gfship._check(gfship.lib().gfship_particles_set_migrate(plist.ptr, C.cast(self._fn, C.c_void_p),
                                                              None))

    def _migrate(self, ctx, nsend, send, nrecv, recv):
        try:
            import gfship
            rs = gfship.lib().gfship_particles_record_size(self.pl.ptr)   # 7, particulates 15
            out = {}
            for d in self.tr.grid.external_sides():
                n = nsend[d]
                out[d] = (np.ctypeslib.as_array(send[d], shape=(n, rs)).copy() if n
                          else np.empty((0, rs)))
            inc = self.tr.exchange_records(out, rs)
            self._keep = {}
            for d in range(6):
                a = inc.get(d)
                if a is None or len(a) == 0:
                    nrecv[d] = 0
                    continue
                a = np.ascontiguousarray(a, dtype=np.float64)
                self._keep[d] = a
                nrecv[d] = len(a)
                recv[d] = a.ctypes.data_as(C.POINTER(C.c_double))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1
