"""The quadtree oracle (oracle/go_tree2d.c: the coarse-fine stencils of SURVEY.md 8f-4) against the
reference's own golden files of test/periodic: r1.ref and r2.ref (one / two extra levels inside the
square |x|, |y| < 0.25), every printed digit, and r0.ref through the uniform tree, where it must
also give the bits of the uniform oracle (go_timestep.c)."""
import os

import numpy as np
import pytest

from oracle import oracle as O


def _rows(golden_dir, name):
    with open(os.path.join(golden_dir, "reference", name)) as f:
        return [l.split() for l in f if l.strip()]


def _run(level, box):
    s = O.Tree2D(periodic=(level, box))
    s.run()
    _, second, infty = s.error_norm()
    return s, ["%.3e" % second, "%.3e" % infty]


def test_tree_shape_box2():
    """the 2:1 constraints of ftt.c:45-83 and the corner rule of ftt.c:2013-2074: two extra levels
    inside the square make a ring of cells one level up around it, corners included"""
    s = O.Tree2D(periodic=(5, 2))
    assert s.depth == 7
    leaves = [int(np.sum(s.flags(l)[1:-1, 1:-1] == 1)) for l in range(s.depth + 1)]
    # 16 x 16 cells of level 5 refined twice; the ring of 18^2 - 16^2 = 68 cells refined once
    assert leaves[7] == 16 * 16 * 16 and leaves[6] == 68 * 4 and leaves[5] == 1024 - 18 * 18
    assert leaves[:5] == [0] * 5
    s.destroy()


@pytest.mark.parametrize("box,levels", [(0, (5, 6)), (1, (5, 6)), (2, (5,))])
def test_periodic_refined_patch_matches_ref(golden_dir, box, levels):
    """test/periodic/periodic.sh: L2 and Linf error of U at t = 0.5 as printed by OutputErrorNorm,
    against r0.ref / r1.ref / r2.ref (the level-7 rows and BOX = 2 at level 6 take minutes on one
    core: tools/tree2d_ref_rows.py prints them)"""
    rows = {int(r[0]): r[1:3] for r in _rows(golden_dir, "periodic_r%d.ref" % box)}
    for level in levels:
        s, got = _run(level, box)
        assert got == rows[level], (box, level, got, rows[level])
        s.destroy()


def test_uniform_tree_equals_uniform_oracle():
    """BOX = 0: the tree code runs the same statements as go_timestep.c on a uniform grid"""
    from flow_cases import oracle_periodic
    level = 5
    a = O.Tree2D(periodic=(level, 0))
    b = oracle_periodic(level)
    a.start()
    b.start()
    for _ in range(8):
        a.step()
        b.step()
    assert a.t == b.t and a.dt == b.advection_params.dt
    for which, f in ((O.Tree2D.U, b.u[0]), (O.Tree2D.V, b.u[1]), (O.Tree2D.P, b.p)):
        assert np.array_equal(a.values(which, level)[1:-1, 1:-1], f.interior())
    a.destroy()
