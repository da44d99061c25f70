"""Test infrastructure for domains of several GfsBoxes (one box per rank / GPU):

  * OracleHooks: the exchange / reduce hooks of the CPU oracle over a transport (numpy views);
  * LocalFabric / LocalTransport: an in-process transport for several boxes of one process (one
    thread per box), used to run several oracle boxes, or several device boxes on a single GPU
    (RCCL refuses two ranks on one device), through the GFSHIP_SIDE_EXTERNAL path;
  * global fields: every box holds ITS part of a field defined on the whole lattice, so that a
    swapped peer, a wrong rank_of or a left/right mix-up changes the results;
  * run_boxes: the thread harness.

Nothing here is product code; the product's own transport is csrc/transport.hip.
"""
import ctypes as C
import threading

import numpy as np

from gfship import distributed as D

GO_EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)
GO_REDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)

def layer_slices(dim, n, side, ghost):
    """numpy index of the interior layer (ghost=False) or ghost layer (ghost=True) along `side`
    of an (n+2)^dim array indexed [k, j, i]"""
    sl = [slice(1, n + 1)] * dim
    axis = dim - 1 - side // 2
    if ghost:
        sl[axis] = 0 if side & 1 else n + 1
    else:
        sl[axis] = 1 if side & 1 else n
    return tuple(sl)


class OracleHooks:
    def __init__(self, oracle_lib, dom_ptr, dim, transport):
        self.dim, self.tr = dim, transport
        self._ex = GO_EXCHANGE_FN(self._exchange)
        self._red = GO_REDUCE_FN(self._reduce)
        oracle_lib.go_domain_set_hooks.restype = None
        oracle_lib.go_domain_set_hooks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]
        oracle_lib.go_domain_set_hooks(dom_ptr, C.cast(self._ex, C.c_void_p), None,
                                       C.cast(self._red, C.c_void_p), None)

    def _exchange(self, ctx, a, level, kind):
        torch = self.tr.torch
        n = 1 << level
        arr = np.ctypeslib.as_array(a, shape=(n + 2,) * self.dim)
        sides = D._kind_sides(self.tr.grid, kind)
        if not sides:
            return
        nface = n ** (self.dim - 1)
        snd, rcv = self.tr.buffers("cpu", nface)
        for s in sides:
            # tangential order: first tangential axis fastest = C order of the [k, j, i] slice
            snd[s].copy_(torch.from_numpy(np.ascontiguousarray(
                arr[layer_slices(self.dim, n, s, False)]).ravel()))
        recv_sides = [x ^ 1 for x in sides] if kind else sides
        self.tr.exchange(sides, snd, recv_sides, rcv)
        for s in recv_sides:
            sl = layer_slices(self.dim, n, s, True)
            arr[sl] = rcv[s].numpy().reshape(arr[sl].shape)

    def _reduce(self, ctx, vals, n, op):
        a = np.ctypeslib.as_array(vals, shape=(n,))
        a[...] = self.tr.allreduce(a.copy(), op)


# ---------------------------------------------------------------------------------------------
# in-process transport: several boxes of one process (threads), used to exercise the
# GFSHIP_SIDE_EXTERNAL path of libgfship on a single GPU
# ---------------------------------------------------------------------------------------------

_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        import ctypes as C
        _HIP = C.CDLL("libamdhip64.so")
    return _HIP


class LocalFabric:
    """shared state of the LocalTransports of one process"""

    def __init__(self, nboxes):
        import threading
        self.n = nboxes
        self.barrier = threading.Barrier(nboxes)
        self.posted = [None] * nboxes
        self.values = [None] * nboxes


class LocalTransport:
    def __init__(self, grid, rank, fabric, device=None):
        import torch
        self.torch = torch
        self.grid, self.rank, self.fabric = grid, rank, fabric
        self.device = device if device is not None else torch.device("cpu")
        self._bufs = {}
        self.stream = None      # set by DeviceHooks users that want stream synchronisation

    buffers = D.Transport.buffers

    def exchange(self, send_sides, snd, recv_sides, rcv):
        f = self.fabric
        if self.device.type == "cuda":
            self.torch.cuda.current_stream().synchronize()     # my packs are complete
        f.posted[self.rank] = {s: snd[s] for s in send_sides}
        f.barrier.wait()
        for r in recv_sides:
            peer = self.grid.neighbour(self.rank, r)
            rcv[r].copy_(f.posted[peer][r ^ 1])
        if self.device.type == "cuda":
            self.torch.cuda.current_stream().synchronize()     # my copies are complete
        f.barrier.wait()

    def gather_device(self, send, recv, nbytes, stream):
        """MPI_Allgather of device memory between the boxes of this process (the gather hook of
        include/gfship.h): device-to-device copies on the caller's stream"""
        import ctypes as C
        f = self.fabric
        hip = _hip()
        if hip.hipStreamSynchronize(C.c_void_p(stream)):       # what I send is complete
            raise RuntimeError("hipStreamSynchronize")
        f.posted[self.rank] = send
        f.barrier.wait()
        for r in range(f.n):
            if hip.hipMemcpyAsync(C.c_void_p(recv + r * nbytes), C.c_void_p(f.posted[r]), C.c_size_t(nbytes),
                                  3, C.c_void_p(stream)):
                raise RuntimeError("hipMemcpyAsync")
        if hip.hipStreamSynchronize(C.c_void_p(stream)):       # nobody overwrites what I still read
            raise RuntimeError("hipStreamSynchronize")
        f.barrier.wait()

    def exchange_records(self, out, rs=7):
        f = self.fabric
        f.posted[self.rank] = out
        f.barrier.wait()
        inc = {}
        for r in self.grid.external_sides():
            peer = self.grid.neighbour(self.rank, r)
            a = f.posted[peer].get(r ^ 1)
            if a is not None and len(a):
                inc[r] = np.array(a, copy=True)
        f.barrier.wait()
        return inc

    def allreduce(self, vals, op):
        f = self.fabric
        f.values[self.rank] = np.array(vals, dtype=np.float64)
        f.barrier.wait()
        stack = np.stack(f.values)
        out = (stack.sum(0), stack.max(0), stack.min(0))[op]
        f.barrier.wait()
        return out


# ---------------------------------------------------------------------------------------------
# fields on the whole lattice
# ---------------------------------------------------------------------------------------------

def global_centres(grid, rank, n, dim=3):
    """global coordinates in [0, 1)^dim of the cell centres of box `rank` (arrays broadcastable to
    the [k, j, i] interior of the box): box (cx, cy, cz) of a bx x by x bz lattice covers
    [cx/bx, (cx+1)/bx) x ..."""
    c = (np.arange(n) + 0.5) / n
    cc = grid.coords(rank)
    X = (cc[0] + c) / grid.b[0]
    Y = (cc[1] + c) / grid.b[1]
    if dim == 2:
        return X[None, :], Y[:, None]
    Z = (cc[2] + c) / grid.b[2]
    return X[None, None, :], Y[None, :, None], Z[:, None, None]


def lattice_velocity(X, Y, Z):
    """a smooth velocity field with ONE period over the whole lattice and no symmetry between the
    boxes (every box differs from every other and from its own mirror images)"""
    tp = 2. * np.pi
    u = np.sin(tp * X) * np.cos(tp * Y) * np.cos(tp * Z) + 0.3 * np.cos(tp * (Y + 0.1)) + 0.25
    v = -np.cos(tp * X) * np.sin(tp * Y) * np.cos(tp * Z) + 0.2 * np.sin(tp * (Z + 0.3) + tp * X)
    w = 0.15 * np.sin(tp * (X + 0.2)) * np.cos(tp * (Y - 0.15)) + 0.1 * np.cos(tp * Z) + 0. * (X + Y + Z)
    return u + 0. * w, v + 0. * w, w


def assemble(grid, parts, n, dim=3):
    """global [k, j, i] array from the interiors of the boxes, parts[rank]"""
    b = grid.b
    G = np.empty((b[2] * n, b[1] * n, b[0] * n)) if dim == 3 else np.empty((b[1] * n, b[0] * n))
    for r in range(grid.n):
        cx, cy, cz = grid.coords(r)
        if dim == 3:
            G[cz * n:(cz + 1) * n, cy * n:(cy + 1) * n, cx * n:(cx + 1) * n] = parts[r]
        else:
            G[cy * n:(cy + 1) * n, cx * n:(cx + 1) * n] = parts[r]
    return G


def run_boxes(nboxes, worker, timeout=600):
    """worker(rank, fabric) in one thread per box; returns the list of results, raises the first
    exception (a failing box aborts the others' barriers)"""
    fabric = LocalFabric(nboxes)
    out, errors = [None] * nboxes, []

    def body(rank):
        try:
            out[rank] = worker(rank, fabric)
        except BaseException as e:       # noqa: BLE001
            errors.append((rank, e))
            fabric.barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(nboxes)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=timeout)
    real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)]
    if real or errors:
        raise (real or errors)[0][1]
    assert all(not t.is_alive() for t in th), "a box did not finish"
    return out
