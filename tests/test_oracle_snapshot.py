"""The oracle's restatement of the binary cell data of a simulation file (ftt_cell_write_binary +
gfs_cell_write_binary, src/ftt.c:1771-1799, src/domain.c:3176-3207) and of the restart branch of
simulation_run (src/simulation.c:474-475, gfs_update_gradients).  The reference ships no binary
snapshot among its test files, so the byte layout is pinned on its writer's source alone: record
structure, and the order of the leaves, which must be the traversal order that the golden residuals
of test/poisson already pin (tests/test_oracle_golden_poisson.py)."""
import struct

import numpy as np
import pytest

from flow_cases import oracle_reynolds, oracle_taylor_green
from oracle import oracle as O


@pytest.mark.parametrize("dim,level", [(2, 3), (3, 2)])
def test_tree_records_follow_the_traversal_order(dim, level):
    dom = O.Domain(dim, level, [O.SIDE_PERIODIC] * 6)
    rng = np.random.default_rng(3)
    fields = [dom.field(), dom.field()]
    for f in fields:
        for l in range(level + 1):
            f.level(l)[...] = rng.standard_normal(f.level(l).shape)
    data = dom.snapshot_tree(fields)
    rec = 4 + 8 + 8 * len(fields)
    C = 8 if dim == 3 else 4
    ncells = sum(C ** l for l in range(level + 1))
    assert len(data) == ncells * rec
    # walk the records: pre-order, LEAF flag exactly on the deepest level, child ids 0..C-1 in turn
    pos = 0
    leaves = []

    def walk(lv, expect_id):
        nonlocal pos
        flags, marker = struct.unpack_from("<Id", data, pos)
        vals = struct.unpack_from("<%dd" % len(fields), data, pos + 12)
        pos += rec
        assert flags & 7 == expect_id and marker == -1.
        assert bool(flags & 16) == (lv == level)
        if lv == level:
            leaves.append(vals)
        else:
            for n in range(C):
                walk(lv + 1, n)

    walk(0, 0)
    assert pos == len(data)
    # the leaves come in the traversal order of the level (ftt_cell_traverse, pre-order)
    order = dom.order(level)
    for v, f in enumerate(fields):
        flat = f.level(level).ravel()
        assert np.array_equal(np.array([x[v] for x in leaves]), flat[order])


@pytest.mark.parametrize("make,nsteps", [(lambda: oracle_reynolds(4), 3), (lambda: oracle_taylor_green(3), 2)])
def test_restart_from_a_snapshot_continues_bit_for_bit(make, nsteps):
    """run 2n steps = run n, keep what a snapshot keeps (P, Pmac, U, V, W on the leaves, t, i),
    start a fresh simulation from it (time.i > 0: gfs_update_gradients instead of the initial
    projection), run n more"""
    a = make()
    a.start()
    for _ in range(nsteps):
        a.step()
    keep = dict(p=a.p.interior().copy(), pmac=a.pmac.interior().copy(),
                u=[f.interior().copy() for f in a.u], t=a.t, i=a.i)
    for _ in range(nsteps):
        a.step()
    b = make()
    b.p.interior()[...] = keep["p"]
    b.pmac.interior()[...] = keep["pmac"]
    for c, f in enumerate(b.u):
        f.interior()[...] = keep["u"][c]
    b.restart(keep["t"], keep["i"])
    b.start()
    for _ in range(nsteps):
        b.step()
    assert b.t == a.t and b.i == a.i and b.dt == a.dt
    for c in range(a.dim):
        assert np.array_equal(a.u[c].interior(), b.u[c].interior())
    assert np.array_equal(a.p.interior(), b.p.interior())
    assert np.array_equal(a.pmac.interior(), b.pmac.interior())
