"""The reference's test drivers end to end on the device: test/periodic/periodic.sh (nine runs:
r = 0, 1, 2 extra levels inside the square at LEVEL = 5, 6, 7) and test/reynolds/box (`sh ../reynolds.sh
box.gfs 4': LEVEL = 5, 6, 7) with the reference's simulation files, unmodified, through bin/gfship2D,
their outputs compared with `diff' against the reference's golden files r0.ref, r1.ref, r2.ref,
div5.ref, div6.ref, div7.ref (1287 rows) and reynolds.ref.  tools/periodic_rows.sh and
tools/reynolds_box_rows.sh are those drivers (the gnuplot / LaTeX parts left out); about 100 s."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(script, out):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", script), str(out)], capture_output=True,
                       text=True, timeout=900, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def test_periodic_sh_outputs_are_the_reference_files(tmp_path):
    out = _run("periodic_rows.sh", tmp_path / "periodic")
    for r in (0, 1, 2):
        assert "r%d: identical to r%d.ref" % (r, r) in out


def test_reynolds_box_outputs_are_the_reference_files(tmp_path):
    out = _run("reynolds_box_rows.sh", tmp_path / "box")
    for level in (5, 6, 7):
        assert "div%d: identical to div%d.ref" % (level, level) in out
    got = [l.split() for l in open(str(tmp_path / "box" / "reynolds"))]
    want = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "reference",
                                                 "reynolds_box_reynolds.ref"))]
    assert got == want
