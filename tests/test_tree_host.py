"""Host logic of the refined-tree path (csrc/tree.hip) without a device: gfship_tree_host_check builds
the tree, the lists, the dependency levels, the compiled stencils and the plans of the relax loops on
the host and runs every relax loop three ways there -- the reference's program through the stencil
code that walks the tree (ghost copies, cells in tree order, sweep after sweep), the compiled stencils
by dependency level with every level BACKWARDS, the plan of the whole loop with every level backwards,
the flow plan (csrc/tree_flow.hpp: fixed-format micro-operations, results forwarded through a ring of
buffers, old values loaded a level ahead, stores a level late, the branch-free arithmetic wherever the 64
operations of a wavefront are of one kind) with the kernel's timing of loads and stores -- and counts the
values that differ (bit for bit) plus the hazards of the flow plan (a load that a store of the levels
around it could race with).  A wrong read set, a missing dependency or a wrong coefficient in a compiled
stencil shows up as a non-zero count."""
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


@pytest.mark.parametrize("width", [64, 192])
def test_flow_plans_with_fewer_operations_per_level(width):
    """GFSHIP_FLOW_WIDTH (read once per process): the flow plans of a quadtree and of an octree with levels
    of 64 / 192 operations at most: more levels, chains of CHILD -> SUM -> CELL squeezed by the capacity"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import gfship\n"
            "inside = lambda *a: all(-0.25 <= v <= 0.25 for v in a)\n"
            "r = gfship.tree_host_check(lambda x, y: 6 if inside(x, y) else 4)\n"
            "assert r[0] > 0 and r[3] == 0, r\n"
            "r = gfship.tree_host_check(lambda x, y, z: 4 if inside(x, y, z) else 3, dim=3)\n"
            "assert r[0] > 0 and r[3] == 0, r\n")
    env = dict(os.environ)
    env["GFSHIP_FLOW_WIDTH"] = str(width)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "gerris-fft-particles_amd"), env.get("PYTHONPATH", "")])
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
