"""The .gfs front end (gerris-fft-particles_amd/bin/gfship2D, built by csrc/build.sh).

CPU part: the reader and the function compiler (`--check`: parse, compile the C expressions
with the system compiler, describe the run; no device is touched).
GPU part: the repo's own case files (tests/cases/*.gfs: the set-ups of the reference's
test/poisson, test/lid, test/reynolds and test/periodic, written for this repo) run end to end on
the device and are checked against the reference's golden files exactly the way the reference's
test scripts do (poisson.sh, lid.sh, reynolds.sh, periodic.sh)."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "gerris-fft-particles_amd", "bin", "gfship2D")
CASES = os.path.join(ROOT, "tests", "cases")


def _run(case, defs, cwd=None, check=False, exe=BIN):
    cmd = [exe] + (["--check"] if check else []) + ["-D%s=%s" % kv for kv in defs.items()] + \
        [os.path.join(CASES, case)]
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout


def _rows(golden_dir, name):
    return [l.split() for l in open(os.path.join(golden_dir, "reference", name)) if l.strip()]


# ---------------------------------------------------------------------------------------------
# CPU: reader + function compiler
# ---------------------------------------------------------------------------------------------

def test_front_end_is_built():
    assert os.path.exists(BIN), "run __graft_entry__.build()"
    assert os.path.exists(BIN.replace("2D", "3D"))


def test_check_poisson_case_functions():
    out = _run("dirichlet_multigrid.gfs", {"LEVEL": 5, "CYCLE": 3}, check=True)
    assert "class GfsPoisson dim 2 level 5" in out
    assert "iend 1" in out
    x, y = 0.125, -0.25
    div = -math.pi * math.pi * 18 * math.sin(3 * math.pi * x) * math.sin(3 * math.pi * y)
    line = [l for l in out.splitlines() if l.startswith("init Div")][0]
    assert float(line.split("=")[1]) == pytest.approx(div, rel=1e-15)
    bcs = [l.split() for l in out.splitlines() if l.startswith("bc ")]
    assert sorted(b[1] for b in bcs) == ["bottom", "left", "right", "top"]
    assert all(b[2] == "P" and b[3] == "dirichlet" for b in bcs)
    assert float(bcs[0][4]) == pytest.approx(math.sin(3 * math.pi * x) * math.sin(3 * math.pi * y),
                                             rel=1e-15)
    ev = [l.split()[1] for l in out.splitlines() if l.startswith("event ")]
    assert ev == ["OutputProjectionStats", "OutputErrorNorm"]


def test_check_cavity_and_periodic_cases(golden_dir):
    g = os.path.join(golden_dir, "reference")
    out = _run("cavity.gfs", {"LEVEL": 6, "XPROFILE": os.path.join(g, "lid_xprofile"),
                              "YPROFILE": os.path.join(g, "lid_yprofile")}, check=True)
    assert "viscosity 0 0.001" in out and "viscosity 1 0.001" in out
    assert "bc top U dirichlet 1" in out
    assert "event EventStop line" in out and "istep 10" in out
    out = _run("vortex_decay.gfs", {"LEVEL": 5}, check=True)
    assert "right=periodic left=periodic top=periodic bottom=periodic" in out
    assert "init U = <function of variables>" in out       # U = U0: a variable name
    out = _run("vortex_translation.gfs", {"LEVEL": 5}, check=True)
    assert "end 0.5" in out


def test_unsupported_input_fails_loudly(tmp_path):
    bad = tmp_path / "bad.gfs"
    # a refined tree outside the classes that know it (GfsSimulation, GfsPoisson: DESIGN.md 10)
    bad.write_text("1 0 GfsAdvection GfsBox GfsGEdge {} {\n  Refine (x > 0 ? 5 : 4)\n}\nGfsBox {}\n")
    r = subprocess.run([BIN, "--check", str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "non-uniform" in r.stderr and "line 2" in r.stderr
    bad.write_text("1 0 GfsSimulation GfsBox GfsGEdge {} {\n  Refine 4\n  Solid (x*x + y*y - 0.1)\n}\nGfsBox {}\n")
    r = subprocess.run([BIN, "--check", str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "unsupported object `Solid'" in r.stderr


# ---------------------------------------------------------------------------------------------
# GPU: the cases end to end, checked like the reference's test scripts
# ---------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_poisson_case_matches_res7_and_error_refs(golden_dir):
    """test/poisson/poisson.sh: residual after CYCLE V-cycles at level 8 (res-7.ref column 3) and
    error norms after 10 cycles for levels 3..8 (error.ref), read from OutputProjectionStats and
    OutputErrorNorm"""
    for row in _rows(golden_dir, "poisson_res-7.ref"):
        cyc = int(row[0])
        out = _run("dirichlet_multigrid.gfs", {"LEVEL": 8, "CYCLE": cyc})
        res = [l.split() for l in out.splitlines() if l.split()[:1] == ["residual.infty:"]]
        # awk '{if ($1 == "residual.infty:") print CYCLE, $3, $4;}': after, rate
        assert res[-1][2] == row[2], (cyc, res, row)
    for row in _rows(golden_dir, "poisson_error.ref"):
        out = _run("dirichlet_multigrid.gfs", {"LEVEL": int(row[0]), "CYCLE": 10})
        err = [l.split() for l in out.splitlines() if l.startswith("P time:")][-1]
        # awk '{print LEVEL, $5, $7, $9}'
        assert [err[4], err[6], err[8]] == row[1:4], (row, err)


@pytest.mark.gpu
def test_cavity_case_meets_ghia(golden_dir, tmp_path):
    g = os.path.join(golden_dir, "reference")
    _run("cavity.gfs", {"LEVEL": 6, "XPROFILE": os.path.join(g, "lid_xprofile"),
                        "YPROFILE": os.path.join(g, "lid_yprofile")}, cwd=str(tmp_path))
    xp = np.loadtxt(tmp_path / "xprof")
    yp = np.loadtxt(tmp_path / "yprof")
    assert xp.shape == (101, 9) and yp.shape == (101, 9)      # t x y z P Pmac U V DU
    gx = np.loadtxt(os.path.join(g, "xprof.ghia"))
    gy = np.loadtxt(os.path.join(g, "yprof.ghia"))
    # lid.sh: Curve('xprof',3,7) - Curve('xprof.ghia',1,2), Curve('yprof',2,8) - ...
    ex = np.abs(np.interp(gx[:, 0], xp[:, 2], xp[:, 6]) - gx[:, 1]).max()
    ey = np.abs(np.interp(gy[:, 0], yp[:, 1], yp[:, 7]) - gy[:, 1]).max()
    assert ex <= 2e-2 and ey <= 1.7e-2, (ex, ey)
    du = open(tmp_path / "du").read().splitlines()
    assert du[-1].startswith("DU time:") and float(du[-1].split()[-1]) <= 1e-4


@pytest.mark.gpu
def test_vortex_decay_case_matches_div5_and_reynolds_refs(golden_dir, tmp_path):
    """test/reynolds/reynolds.sh at level 5: the Divergence history equals div5.ref line by line,
    and the effective Reynolds number from the kinetic energy file matches reynolds.ref"""
    _run("vortex_decay.gfs", {"LEVEL": 5}, cwd=str(tmp_path))
    got = open(tmp_path / "div").read().splitlines()
    ref = open(os.path.join(golden_dir, "reference", "reynolds_div5.ref")).read().splitlines()
    assert len(got) == len(ref)
    assert got[0].split()[:3] == ref[0].split()[:3]       # t = 0 is round-off noise
    assert got[1:] == ref[1:]
    kin = [l.split() for l in open(tmp_path / "kinetic")]
    ke0, ke1, t1 = float(kin[0][-1]), float(kin[-1][-1]), float(kin[-1][2])
    a = -math.log(ke1 / ke0) / t1
    nu = a / (4. * (2. * 3.14159265359) ** 2)
    rey = {r[0]: float(r[1]) for r in _rows(golden_dir, "reynolds_reynolds.ref")}
    assert 1. / nu == pytest.approx(rey["5"], rel=2e-5)


@pytest.mark.gpu
def test_vortex_translation_case_matches_r0_ref(golden_dir):
    for row in _rows(golden_dir, "periodic_r0.ref")[:2]:
        out = _run("vortex_translation.gfs", {"LEVEL": int(row[0])})
        err = [l.split() for l in out.splitlines() if l.startswith("U time:")][-1]
        # periodic.sh: awk '{print LEVEL, $7, $9}': second, infty
        assert [err[6], err[8]] == row[1:3], (row, err)


# ---------------------------------------------------------------------------------------------
# GfsParticleList of GfsParticulate objects with forces (modules/particulates)
# ---------------------------------------------------------------------------------------------

def test_check_particulates_case():
    out = _run("particulates.gfs", {"LEVEL": 5, "NSTEPS": 8}, check=True)
    ev = [l.split()[1] for l in out.splitlines() if l.startswith("event ")]
    assert ev == ["ParticleList", "OutputTime"]
    assert "viscosity 0 0.001" in out and "iend 8" in out


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["particulates.gfs", "particulates_coeff.gfs"])
def test_particulates_case_matches_the_oracle(tmp_path, case):
    """the list of the case file after 8 steps, as the reference would write it (%g), against the
    oracle driven from Python with the same set-up (previous velocity of the GfsForceCoeff objects
    stored while the fields are still zero, like the reference reading the file); the second file
    gives two of the forces a GfsFunction coefficient (compiled for the device by the front end's
    library call, evaluated by the oracle as Python callbacks)"""
    import math
    from oracle import oracle as O
    from flow_cases import PERIODIC, reynolds_init
    level, nsteps = 5, 8
    outp = tmp_path / "plist.txt"
    cmd = [BIN, "--particles", str(outp), "-DLEVEL=%d" % level, "-DNSTEPS=%d" % nsteps,
           os.path.join(CASES, case)]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in open(outp) if l.strip().startswith("GfsParticulate")]
    got = np.array([[float(x) for x in row[1:]] for row in rows])

    s = O.Sim(2, level, PERIODIC)
    for c in range(2):
        s.set_viscosity(c, 1e-3)
    s.approx_projection_params.tolerance = 1e-6
    s.projection_params.tolerance = 1e-6
    s.set_time(end=2.)
    pos = np.array([[0.11, 0.23, 0], [-0.31, 0.07, 0], [0.4, -0.4, 0], [-0.05, -0.27, 0], [0.25, 0.25, 0]])
    vel = np.array([[0, 0, 0], [0.1, -0.2, 0], [0, 0.3, 0], [-0.1, 0, 0], [0, 0, 0]], dtype=float)
    mass = np.array([2e-3, 1.5e-3, 0.5e-3, 3e-3, 1e-3])
    vol = np.array([1e-3, 1e-3, 1e-3, 2e-3, 0.5e-3])
    pl = O.Particles(s, pos, np.arange(1, 6, dtype=np.uint32))
    pl.set_particulate(vel, mass, vol)
    pl.set_forces([O.FORCE_DRAG, O.FORCE_LIFT, O.FORCE_INERTIAL])      # Un = Vn = 0 here
    if case == "particulates_coeff.gfs":
        pl.set_coefficient(0, lambda rep, u, v, w, d: 24. / rep * (1. + 0.15 * math.pow(rep, 0.687)))
        pl.set_coefficient(1, lambda rep, u, v, w, d: 0.3 + 0.1 * u)
    x, y = s.dom.centres()
    u, v = reynolds_init(x, y)
    s.u[0].interior()[...] = u
    s.u[1].interior()[...] = v
    s.start()
    for k in range(nsteps):
        pl.event()
        s.step()
    pl.event()
    op, oi = pl.state()
    ov, om, of = pl.particulate_state()
    assert [int(g) for g in got[:, 0]] == list(oi)
    idx = np.array(oi) - 1
    want = np.column_stack([oi, op, om, vol[idx], ov, of])
    assert np.allclose(got, want, rtol=2e-5, atol=1e-9)


# ---------------------------------------------------------------------------------------------
# GfsOutputEnergySpectra (modules/fft)
# ---------------------------------------------------------------------------------------------

def test_check_spectra_case():
    out = _run("spectra.gfs", {"LEVEL": 5, "NSTEPS": 3}, check=True, exe=BIN.replace("2D", "3D"))
    ev = [l.split()[1] for l in out.splitlines() if l.startswith("event ")]
    assert ev == ["OutputEnergySpectra", "OutputEnergySpectra"]


@pytest.mark.gpu
def test_spectra_case_taylor_green(tmp_path):
    """the spectrum files in the reference's format: all the energy of the initial field in
    |k|^2 = 3, the total equal to the kinetic energy; three steps later it is still there (the
    flow has barely evolved) and nothing sits beyond the grid's resolved shells"""
    level = 5
    r = subprocess.run([BIN.replace("2D", "3D"), "-DLEVEL=%d" % level, "-DNSTEPS=3",
                        os.path.join(CASES, "spectra.gfs")], cwd=tmp_path, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    n = 1 << level
    for name in ("spectrum-0", "spectrum-end"):
        lines = open(tmp_path / name).read().splitlines()
        assert lines[0].startswith("# Total energy = ") and lines[1].strip() == "# 1:k 2:Ek"
        etot = float(lines[0].split("=")[1])
        rows = np.array([[float(x) for x in l.split()] for l in lines[2:]])
        assert len(rows) == 4 * (n // 2 + 1) ** 2 - 1
        dk = 2 * np.pi / ((n - 1) / n)
        assert np.isclose(rows[2, 0], dk * np.sqrt(3.), rtol=1e-5)
        x = (np.arange(n) + 0.5) / n - 0.5
        Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
        ke = 0.5 * 2 * np.mean((np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z)) ** 2)
        if name == "spectrum-0":
            assert np.isclose(etot, ke, rtol=1e-5)
            assert np.isclose(rows[2, 1], ke, rtol=1e-5)
        else:
            assert 0.9 * ke < etot <= ke * (1 + 1e-6)
            assert rows[2, 1] > 0.9 * etot


@pytest.mark.gpu
def test_output_spectra_and_turbulent_viscosity_case(tmp_path):
    """GfsOutputSpectra in write_spectra's format (modules/fft.c:1047-1085): the Taylor-Green U has
    eight modes (+-1, +-1, +1) of amplitude 1/8; GfsVariableTurbulentViscosity is refreshed every step
    and its norms are those of (Cs h)^2 |S| of the same field"""
    level = 4
    r = subprocess.run([BIN.replace("2D", "3D"), "-DLEVEL=%d" % level, "-DNSTEPS=2",
                        os.path.join(CASES, "spectra_variable.gfs")], cwd=tmp_path, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    n = 1 << level
    lines = open(tmp_path / "spectra-U").read().splitlines()
    assert lines[0].strip() == "# %d" % n ** 3 and lines[1] == "# 1:kx 2:ky 3:kz 4:real 5:img"
    rows = np.array([[float(x) for x in l.split()] for l in lines[2:]])
    assert len(rows) == n * n * (n // 2 + 1)
    dk = 2 * np.pi / ((n - 1) / n)
    amp = np.hypot(rows[:, 3], rows[:, 4])
    big = rows[amp > 0.05]
    assert len(big) == 4                                   # kz >= 0 half: (+-1, +-1, 1)
    assert np.allclose(np.abs(big[:, :3]), dk, rtol=1e-5)
    assert np.allclose(np.hypot(big[:, 3], big[:, 4]), 0.125, rtol=1e-3)
    # the plane z = 0.1 (a flat box: realdim == 2): N x (N/2 + 1) rows, kz = 0, the mode (+-1, 1) of
    # sin (2 pi x) cos (2 pi y) with the amplitude cos (2 pi z_cell)/4
    lines = open(tmp_path / "spectra-U-plane").read().splitlines()
    assert lines[0].strip() == "# %d" % n ** 2
    prow = np.array([[float(x) for x in l.split()] for l in lines[2:]])
    assert len(prow) == n * (n // 2 + 1) and np.all(prow[:, 2] == 0.)
    pamp = np.hypot(prow[:, 3], prow[:, 4])
    pbig = prow[pamp > 0.05]
    assert len(pbig) == 2 and np.allclose(np.abs(pbig[:, :2]), dk, rtol=1e-5)
    zc = -0.5 + (np.floor((0.1 + 0.5) * n) + 0.5) / n
    assert np.allclose(np.hypot(pbig[:, 3], pbig[:, 4]), abs(np.cos(2 * np.pi * zc)) / 4., rtol=1e-3)
    # the eddy viscosity against the numpy restatement on the initial field
    from oracle.go_spectra import turbulent_viscosity
    x = (np.arange(-1, n + 1) + 0.5) / n - 0.5
    Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
    u = [np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z),
         -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.cos(2 * np.pi * Z), 0. * X]
    nut = turbulent_viscosity(u, 0.17)
    first = open(tmp_path / "nut").read().splitlines()[0].split()
    vals = {k: float(first[first.index(k) + 1]) for k in ("first:", "infty:")}
    assert np.isclose(vals["first:"], np.abs(nut).mean(), rtol=1e-3)
    assert np.isclose(vals["infty:"], np.abs(nut).max(), rtol=1e-3)


@pytest.mark.gpu
def test_isotropic_case_init_spectra(tmp_path):
    """GfsInitSpectra + GfsOutputEnergySpectra from a .gfs file: the synthetic field has about the
    requested energy, a decaying spectrum, and survives three projection / advection steps"""
    level = 5
    r = subprocess.run([BIN.replace("2D", "3D"), "-DLEVEL=%d" % level, "-DNSTEPS=3",
                        os.path.join(CASES, "isotropic.gfs")], cwd=tmp_path, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    e = {}
    for name in ("spectrum-0", "spectrum-end"):
        lines = open(tmp_path / name).read().splitlines()
        e[name] = float(lines[0].split("=")[1])
        rows = np.array([[float(x) for x in l.split()] for l in lines[2:]])
        first = rows[:6, 1]
        assert np.all(first[first > 0][:-1] > first[first > 0][1:])
    # at 32^3 the interpolation onto the cell centres removes about half of the energy (most of it sits
    # near the grid scale with ReL = 1000); 128^3 keeps > 80 % (tests/test_gpu_spectra.py)
    assert 0.4 * 1.5 < e["spectrum-0"] < 1.05 * 1.5
    assert 0.5 * e["spectrum-0"] < e["spectrum-end"] <= e["spectrum-0"]
    div = [l for l in open(tmp_path / "div").read().splitlines() if l.strip()]
    assert len(div) == 4


# ---------------------------------------------------------------------------------------------
# GfsAdvection + GfsVariableStreamFunction (test/advection)
# ---------------------------------------------------------------------------------------------

def test_check_rotation_case():
    out = _run("rotation.gfs", {"LEVEL": 5}, check=True)
    assert "class GfsAdvection dim 2 level 5" in out
    ev = [l.split()[1] for l in out.splitlines() if l.startswith("event ")]
    assert ev == ["OutputErrorNorm", "OutputScalarSum"]


@pytest.mark.gpu
def test_rotation_case_matches_error_ref(golden_dir, tmp_path):
    """test/advection/advection.sh: error norms of the tracer after one revolution for levels 5..7
    against error.ref -- second and maximum norm to the last digit or so of the file, the first norm to 1 %
    (the reference box has no ghost cells at its walls, see tests/test_oracle_golden_timestep.py)
    -- and the tracer sum (OutputScalarSum)"""
    ref = {int(r[0]): r[1:] for r in _rows(golden_dir, "advection_error.ref")}
    for level in (5, 6, 7):
        _run("rotation.gfs", {"LEVEL": level}, cwd=str(tmp_path))
        err = open(tmp_path / ("error-%d" % level)).read().split()
        # awk '{ print LEVEL " " $5 " " $7 " " $9}'
        first, second, infty = err[4], err[6], err[8]
        assert np.allclose([float(second), float(infty)], [float(v) for v in ref[level][1:3]],
                           rtol=2e-3), (level, err, ref[level])
        assert abs(float(first) / float(ref[level][0]) - 1.) < 0.01
        sums = [float(l.split()[4]) for l in open(tmp_path / ("t-%d" % level)) if l.strip()]
        # the rotation crosses the walls of the square box: the sum is only nearly conserved
        assert abs(sums[-1] - sums[0]) < 1e-3 * abs(sums[0])


@pytest.mark.gpu
def test_refined_cube_case_3d():
    """tests/cases/refined_cube.gfs: an octree with two extra levels inside a cube (gfship3D, the
    refined-tree path) against the octree oracle set up by hand with the same numbers: the volume-
    weighted norms OutputScalarNorm prints, to the printed digits, and the time of OutputTime"""
    from oracle import oracle as O
    level, box, nsteps = 3, 2, 3
    out = _run("refined_cube.gfs", {"LEVEL": level, "BOX": box, "NSTEPS": nsteps},
               exe=BIN.replace("2D", "3D"))
    inside = lambda a: not (a < -0.25 or a > 0.25)
    s = O.Tree(refine=lambda x, y, z: level + box if (inside(x) and inside(y) and inside(z)) else level,
                 dim=3)
    for l in range(s.depth + 1):
        x, y, z = s.centres(l)
        s.values(O.Tree.U, l)[...] = np.sin(2. * np.pi * x) * np.cos(2. * np.pi * y) * np.cos(2. * np.pi * z) + 0.3
        s.values(O.Tree.V, l)[...] = - np.cos(2. * np.pi * x) * np.sin(2. * np.pi * y) * np.cos(2. * np.pi * z) - 0.2
        s.values(O.Tree.W, l)[...] = 0.1 * np.sin(2. * np.pi * (x + y)) * np.sin(2. * np.pi * z)
    s.projection_params.tolerance = s.approx_projection_params.tolerance = 1e-5
    s.set_time(1e30, 0.75)
    s.start()
    for _ in range(nsteps):
        s.step()
    lines = out.splitlines()
    for name, which in (("U", O.Tree.U), ("W", O.Tree.W)):
        first = second = wsum = 0.
        infty = 0.
        for l in range(s.depth + 1):
            leaf = s.flags(l)[1:-1, 1:-1, 1:-1] == 1
            if not leaf.any():
                continue
            a = np.abs(s.values(which, l)[1:-1, 1:-1, 1:-1][leaf])
            w = 1. / (1 << l) ** 3
            first += w * float(a.sum())
            second += w * float((a * a).sum())
            wsum += w * a.size
            infty = max(infty, float(a.max()))
        want = "%s time: %g first: % 10.3e second: % 10.3e infty: % 10.3e" % (
            name, s.t, first / wsum, math.sqrt(second / wsum), infty)
        assert want in lines, (want, lines)
    step = [l for l in lines if l.startswith("step:")][0].split()
    assert int(step[1]) == nsteps and float(step[3]) == pytest.approx(s.t, abs=1e-8)
    s.destroy()


# ---------------------------------------------------------------------------------------------
# GfsPhysicalParams { alpha = f (T) }: variable density following a tracer
# ---------------------------------------------------------------------------------------------

def test_check_variable_density_case():
    out = _run("variable_density.gfs", {"LEVEL": 5, "NSTEPS": 4}, check=True)
    assert "iend 4" in out
    bad = "1 0 GfsSimulation GfsBox GfsGEdge {} {\n  Refine 4\n  PhysicalParams { L = 2 }\n}\nGfsBox {}\n"
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".gfs") as f:
        f.write(bad)
        f.flush()
        r = subprocess.run([BIN, "--check", f.name], capture_output=True, text=True)
    assert r.returncode != 0 and "only L = 1" in r.stderr


@pytest.mark.gpu
def test_variable_density_case_matches_the_oracle():
    """tests/cases/variable_density.gfs through gfship2D against the oracle driven from Python: alpha
    evaluated on the faces before every step from the face-interpolated tracer, exactly as
    gfs_function_face_value does; the norms OutputScalarNorm prints, to the printed digits, and the time"""
    from oracle import oracle as O
    from flow_cases import PERIODIC
    level, nsteps = 5, 4
    out = _run("variable_density.gfs", {"LEVEL": level, "NSTEPS": nsteps})
    s = O.Sim(2, level, PERIODIC)
    x, y = s.dom.centres()
    s.u[0].interior()[...] = - np.cos(2. * np.pi * x) * np.sin(2. * np.pi * y)
    s.u[1].interior()[...] = np.sin(2. * np.pi * x) * np.cos(2. * np.pi * y)
    T = s.add_tracer()
    T.interior()[...] = np.exp(- 30. * ((x - 0.1) * (x - 0.1) + (y + 0.05) * (y + 0.05)))
    s.approx_projection_params.tolerance = s.projection_params.tolerance = 1e-6
    alpha = [O.Field(s.dom, -1) for _ in range(2)]
    s.set_alpha(alpha)

    def refresh():
        t = T.leaf()                     # with the ghost layer of the last BC application
        for c in range(2):
            ax = 1 - c
            nb = np.roll(t, -1, axis=ax)                       # the + neighbour (the last entry is unused)
            tf = ((1. - 0.5) * t + 0.5 * nb) / 1.
            alpha[c].leaf()[...] = 1. / (1. + 2. * tf)

    O.lib().go_bc(T.ptr, T.ptr, level)
    refresh()
    s.start()
    for _ in range(nsteps):
        refresh()
        s.step()
    lines = out.splitlines()
    n = 1 << level
    for name, f in (("U", s.u[0]), ("P", s.p), ("T", T)):
        a = np.abs(f.interior())
        want = "%s time: %g first: % 10.3e second: % 10.3e infty: % 10.3e" % (
            name, s.t, a.sum() / n ** 2, math.sqrt((a * a).sum() / n ** 2), a.max())
        assert want in lines, (want, [l for l in lines if l.startswith(name + " time")])


@pytest.mark.gpu
def test_refined_tracer_case_matches_the_tree_oracle():
    """tests/cases/refined_tracer.gfs (a GfsVariableTracer on the refined quadtree of test/periodic) through
    gfship2D against the tree oracle: the volume-weighted norms and the sum OutputScalarNorm / OutputScalarSum
    print, to the printed digits"""
    from oracle import oracle as O
    level, box, nsteps = 4, 2, 5
    out = _run("refined_tracer.gfs", {"LEVEL": level, "BOX": box, "NSTEPS": nsteps})
    s = O.Tree(periodic=(level, box))
    k = s.add_tracer(1)
    for l in range(s.depth + 1):
        x, y = s.centres(l)
        s.values(k, l)[...] = np.exp(- 30. * ((x - 0.2) * (x - 0.2) + (y - 0.2) * (y - 0.2)))
    s.set_time(1e30, 0.75)
    s.start()
    for _ in range(nsteps):
        s.step()
    first = second = wsum = total = 0.
    infty = 0.
    for l in range(s.depth + 1):
        leaf = s.flags(l)[1:-1, 1:-1] == 1
        if not leaf.any():
            continue
        a = s.values(k, l)[1:-1, 1:-1][leaf]
        w = 1. / (1 << l) ** 2
        first += w * float(np.abs(a).sum())
        second += w * float((a * a).sum())
        total += w * float(a.sum())
        wsum += w * a.size
        infty = max(infty, float(np.abs(a).max()))
    lines = out.splitlines()
    want = "T time: %g first: % 10.3e second: % 10.3e infty: % 10.3e" % (
        s.t, first / wsum, math.sqrt(second / wsum), infty)
    assert want in lines, (want, [l for l in lines if l.startswith("T time")])
    got = [l.split() for l in lines if l.startswith("T time") and "sum:" in l][0]
    assert float(got[-1]) == pytest.approx(total, rel=1e-5)


@pytest.mark.gpu
def test_refined_cavity_case_matches_the_tree_oracle():
    """tests/cases/refined_cavity.gfs (test/lid on a quadtree refined near the walls: BcDirichlet on U, V,
    GfsSourceDiffusion) through gfship2D against the tree oracle after NSTEPS steps: the volume-weighted norms
    OutputScalarNorm prints, to the printed digits, and the time"""
    from oracle import oracle as O
    level, nsteps = 4, 25
    out = _run("refined_cavity.gfs", {"LEVEL": level, "NSTEPS": nsteps})
    refine = lambda x, y: level + 1 if (x < -0.25 or x > 0.25 or y < -0.25 or y > 0.25) else level
    s = O.Tree(refine=refine, sides=[O.SIDE_BOUNDARY] * 4)
    for c in range(2):
        for d in range(4):
            s.set_bc_u(c, d, O.BC_DIRICHLET, 1. if (c == 0 and d == 2) else 0.)
        s.set_viscosity(c, 1e-3)
    s.set_time(300., 0.8)
    s.start()
    for _ in range(nsteps):
        s.step()
    lines = out.splitlines()
    for name, which in (("U", O.Tree.U), ("V", O.Tree.V), ("P", O.Tree.P)):
        first = second = wsum = 0.
        infty = 0.
        for l in range(s.depth + 1):
            leaf = s.flags(l)[1:-1, 1:-1] == 1
            if not leaf.any():
                continue
            a = np.abs(s.values(which, l)[1:-1, 1:-1][leaf])
            w = 1. / (1 << l) ** 2
            first += w * float(a.sum())
            second += w * float((a * a).sum())
            wsum += w * a.size
            infty = max(infty, float(a.max()))
        want = "%s time: %g first: % 10.3e second: % 10.3e infty: % 10.3e" % (
            name, s.t, first / wsum, math.sqrt(second / wsum), infty)
        assert want in lines, (want, [l for l in lines if l.startswith(name + " time")])
