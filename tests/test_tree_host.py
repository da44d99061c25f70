"""Host logic of the refined-tree path (csrc/tree.hip) without a device: gfship_tree_host_check builds
the tree, the lists, the dependency levels, the compiled stencils and the plans of the relax loops on
the host and runs every relax loop three ways there -- the reference's program through the stencil
code that walks the tree (ghost copies, cells in tree order, sweep after sweep), the compiled stencils
by dependency level with every level BACKWARDS, the plan of the whole loop with every level backwards
-- and counts the values that differ (bit for bit).  A wrong read set, a missing dependency or a
wrong coefficient in a compiled stencil shows up as a non-zero count."""
import pytest

import gfship

inside = lambda *a: all(-0.25 <= v <= 0.25 for v in a)

CASES = [
    ("quadtree, two extra levels in the square", dict(refine=lambda x, y: 6 if inside(x, y) else 4)),
    ("quadtree, one extra level, 3 sweeps", dict(refine=lambda x, y: 6 if inside(x, y) else 5, nrelax=3)),
    ("uniform quadtree", dict(refine=lambda x, y: 4)),
    ("octree, off-centre ball with two extra levels",
     dict(refine=lambda x, y, z: 5 if (x - 0.1) ** 2 + (y + 0.05) ** 2 + z * z < 0.04 else 3, dim=3)),
    ("octree, cube with one extra level", dict(refine=lambda x, y, z: 4 if inside(x, y, z) else 3, dim=3)),
    ("quadtree with GfsBoundary sides, circle with two extra levels",
     dict(refine=lambda x, y: 6 if x * x + y * y <= 0.0625 else 4, sides=[gfship.SIDE_BOUNDARY] * 4)),
    ("octree with GfsBoundary sides", dict(refine=lambda x, y, z: 4 if x * x + y * y + z * z < 0.05 else 2,
                                           dim=3, sides=[gfship.SIDE_BOUNDARY] * 6)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_plans_of_a_tree_reproduce_the_sequential_program(name, kw):
    updates, levels_sweeps, levels_loop, differ = gfship.tree_host_check(**kw)
    assert updates > 0
    assert differ == 0
    # the plan of a whole loop never needs more levels than the sweeps one after the other (+ the
    # copies of the ghosts between them)
    nrelax = kw.get("nrelax", 4)
    assert levels_loop <= levels_sweeps + 2 * nrelax * 8


def test_refinement_that_differs_across_a_periodic_side_is_refused():
    with pytest.raises(gfship.GfshipError, match="periodic side"):
        gfship.tree_host_check(lambda x, y: 5 if x > 0.25 else 4)
