"""Simulation files that carry cell data (GfsOutputSimulation), SURVEY.md 8f rank 3.

CPU: the comparison tool (gfshipcompare2D/3D, the reference's tools/gfscompare.c for files with
the same tree) on files put together from the oracle's cell data, binary and text, against numpy.
GPU: the front end writes a snapshot in the middle of a run, restarts from it and ends in the very
bytes of the uninterrupted run; the tool reports a zero difference."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O

BIN = os.path.join(ROOT, "gerris-fft-particles_amd", "bin")
CASES = os.path.join(ROOT, "tests", "cases")


def _file(path, dim, dom, fields, names, binary, t=0.5, i=7):
    data = dom.snapshot_tree(fields)
    nleaf = (1 << dom.depth) ** dim
    head = ("# Gerris Flow Solver %dD version 1.3.2 (test)\n"
            "1 %d GfsSimulation GfsBox GfsGEdge { version = 120812 variables = %s %s} {\n"
            "  GfsTime { i = %d t = %.17g }\n}\n"
            "GfsBox { id = 1 pid = -1 size = %d x = 0 y = 0 z = 0 } {\n"
            % (dim, dim, ",".join(names), "binary = 1 " if binary else "", i, t, nleaf)).encode()
    if not binary:
        rec = 12 + 8 * len(fields)
        lines = []
        for o in range(0, len(data), rec):
            flags = int.from_bytes(data[o:o + 4], "little")
            vals = np.frombuffer(data[o + 12:o + rec], dtype="<f8")
            lines.append("%u -1 " % flags + " ".join("%.17g" % v for v in vals))
        data = ("\n".join(lines) + "\n").encode()
    tail = b"}\n" + "".join("1 1 %s\n" % d for d in ("right", "top", "front")[:dim]).encode()
    with open(path, "wb") as f:
        f.write(head + data + tail)


def _compare(dim, *args):
    r = subprocess.run([os.path.join(BIN, "gfshipcompare%dD" % dim)] + list(args),
                       capture_output=True, text=True, timeout=120)
    return r


@pytest.mark.parametrize("dim,level", [(2, 4), (3, 3)])
def test_compare_tool_norms_and_constant_shift(tmp_path, dim, level):
    rng = np.random.default_rng(dim)
    dom = O.Domain(dim, level, [O.SIDE_PERIODIC] * 6)
    P1, U1, P2, U2 = (dom.field() for _ in range(4))
    for f in (P1, U1, P2, U2):
        f.interior()[...] = rng.standard_normal(f.interior().shape)
    P2.interior()[...] = P1.interior() + 3.25 + 1e-3 * rng.standard_normal(P1.interior().shape)
    a, b, c = (str(tmp_path / n) for n in ("a.gfs", "b.gfs", "c.gfs"))
    _file(a, dim, dom, [P1, U1], ["P", "U"], True)
    _file(b, dim, dom, [U2, P2], ["U", "P"], True)      # another column order
    _file(c, dim, dom, [U2, P2], ["U", "P"], False)     # the same as a text tree
    e = P1.interior() - P2.interior()
    for other in (b, c):
        r = _compare(dim, "-v", a, other, "P")
        assert r.returncode == 0, r.stderr
        m = re.search(r"total err first:\s*(\S+) second:\s*(\S+) infty:\s*(\S+) w: (\S+)", r.stderr)
        got = [float(x) for x in m.groups()]
        want = [np.abs(e).mean(), np.sqrt((e * e).mean()), np.abs(e).max(), 1.]
        assert got == pytest.approx(want, rel=2e-3)
        r = _compare(dim, "-v", "-C", a, other, "P")
        ec = e - e.mean()
        m = re.search(r"total err first:\s*(\S+) second:\s*(\S+) infty:\s*(\S+) w: (\S+)", r.stderr)
        got = [float(x) for x in m.groups()]
        assert got == pytest.approx([np.abs(ec).mean(), np.sqrt((ec * ec).mean()), np.abs(ec).max(), 1.],
                                    rel=2e-3)
    assert _compare(dim, a, b, "V").returncode == 1           # unknown variable
    assert "unknown variable" in _compare(dim, a, b, "V").stderr
    # a file of the other dimension is refused (child ids / record count do not fit)
    assert _compare(5 - dim, "-v", a, b, "P").returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("binary", [1, 0])
def test_front_end_restart_from_its_own_snapshot(tmp_path, binary):
    """vortex_decay (test/reynolds set-up): a run of 8 steps that writes the simulation at step 4
    and at the end, and a second run started from the step-4 file.  With a binary tree the restart
    ends in the very same cell data; with a text tree (%g, 6 digits) it cannot, and the tool says
    how far off it is."""
    exe = os.path.join(BIN, "gfship2D")
    case = open(os.path.join(CASES, "vortex_decay.gfs")).read()
    extra = ("  OutputSimulation { istart = 4 istep = 100 } mid.gfs { binary = %d }\n"
             "  OutputSimulation { start = end } end.gfs { binary = 1 }\n" % binary)
    # put the outputs in front of the closing brace of the simulation body
    k = case.rindex("}", 0, case.rindex("GfsBox"))
    text = (case[:k] + extra + case[k:]).replace("Time { end = 2 }", "Time { end = 2 iend = 8 }")
    (tmp_path / "run.gfs").write_text(text)
    r = subprocess.run([exe, "-DLEVEL=5", "run.gfs"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    mid, end = tmp_path / "mid.gfs", tmp_path / "end.gfs"
    assert b"GfsTime { i = 4 " in mid.read_bytes() and b"GfsTime { i = 8 " in end.read_bytes()
    (tmp_path / "first").mkdir()
    os.rename(str(end), str(tmp_path / "first" / "end.gfs"))
    r = subprocess.run([exe, "mid.gfs"], cwd=str(tmp_path), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr
    a, b = (tmp_path / "first" / "end.gfs").read_bytes(), end.read_bytes()
    for var in ("U", "V", "P"):
        c = _compare(2, "-v", str(tmp_path / "first" / "end.gfs"), str(end), var)
        assert c.returncode == 0, c.stderr
        m = re.search(r"total err first:\s*(\S+) second:\s*(\S+) infty:\s*(\S+)", c.stderr)
        err = float(m.group(3))
        if binary:
            assert err == 0.
        else:
            assert 0. < err < 1e-4
    if binary:
        # and the time, to the last digit (the columns of the two files may come in another order:
        # the variables an Init object added are declared by the snapshot's header in the second run)
        assert re.search(rb"GfsTime \{[^}]*\}", a).group(0) == re.search(rb"GfsTime \{[^}]*\}", b).group(0)
