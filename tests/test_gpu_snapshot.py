"""Binary snapshots on the device: the byte image of the cell tree built in
HBM must equal the oracle's walk of the tree byte for byte; reading it back restores every level;
a simulation restarted from a snapshot (time.i > 0) continues bit for bit."""
import numpy as np
import pytest

import gfship
from flow_cases import PERIODIC, oracle_reynolds, oracle_taylor_green
from oracle import oracle as O
from test_gpu_timestep import _assert_same_state, _device_sim

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,level", [(2, 5), (3, 4), (3, 6)])
def test_tree_image_equals_the_oracle_walk_and_reads_back(dim, level):
    od = O.Domain(dim, level, PERIODIC)
    gd = gfship.Domain(dim, level, PERIODIC)
    rng = np.random.default_rng(level)
    ofs, gfs = [], []
    for _ in range(3):
        of, gf = od.field(), gd.variable()
        for l in range(level + 1):
            a = rng.standard_normal(of.level(l).shape)
            of.level(l)[...] = a
            gf.upload(a, l)
        ofs.append(of)
        gfs.append(gf)
    want = od.snapshot_tree(ofs)
    got = gd.snapshot_tree(gfs)
    assert len(got) == len(want)
    assert got == want
    # read it back into fresh variables: every level, interior cells
    back = [gd.variable() for _ in gfs]
    gd.snapshot_tree_read(back, got)
    for of, gf in zip(ofs, back):
        for l in range(level + 1):
            sl = (slice(1, -1),) * dim
            assert np.array_equal(gf.download(l)[sl], of.level(l)[sl]), l
    # a file of another dimension / depth is refused
    with pytest.raises(gfship.GfshipError):
        gd.snapshot_tree_read(back, got[:-20])
    bad = bytearray(got)
    bad[12 + 8 * len(gfs)] ^= 1          # the child id (flags) of the second record
    with pytest.raises(gfship.GfshipError, match="FTT_CELL_ID"):
        gd.snapshot_tree_read(back, bytes(bad))
    gd.destroy()


@pytest.mark.parametrize("make,level,nsteps", [(oracle_reynolds, 5, 3), (oracle_taylor_green, 5, 2)])
def test_restart_from_a_snapshot_continues_bit_for_bit(make, level, nsteps):
    """run 2n steps = run n, write, read into a fresh simulation, run n more (on the device), and
    both equal the uninterrupted oracle"""
    osim = make(level)
    gd, gs = _device_sim(osim, PERIODIC)
    if osim.dim == 2:
        gs.set_time(end=2.)
    osim.start()
    gs.start()
    for _ in range(nsteps):
        osim.step()
        gs.step()
    variables = [gs.p, gs.pmac] + list(gs.u)
    image = gd.snapshot_tree(variables)
    # every level of every variable as the oracle holds it at this point of the loop: the non-leaf
    # values are those of gfs_cell_coarse_init before the last approximate projection (U, V, W: filled
    # by the pass that corrects the leaves; Pmac: computed now, on demand; P: by coarse_init_kernel)
    assert image == osim.dom.snapshot_tree([osim.p, osim.pmac] + list(osim.u[:osim.dim]))
    t, i = gs.t, gs.i
    for _ in range(nsteps):
        osim.step()
        gs.step()
    _assert_same_state(osim, gs, "uninterrupted")
    # fresh domain and simulation from the image
    o2 = make(level)
    gd2, gs2 = _device_sim(o2, PERIODIC)
    if osim.dim == 2:
        gs2.set_time(end=2.)
    gd2.snapshot_tree_read([gs2.p, gs2.pmac] + list(gs2.u), image)
    gs2.restart(t, i)
    gs2.start()
    for _ in range(nsteps):
        gs2.step()
    _assert_same_state(osim, gs2, "restarted")
    assert gs2.i == gs.i
    gd.destroy()
    gd2.destroy()
