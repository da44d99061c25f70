"""GPU parity of the Poisson path: libgfship (through the C ABI) against the CPU oracle on the
same inputs.  Exact mode must be bit-identical (np.array_equal) for fields and for the
max-norm; the summed norms are tree-reduced on the device and agree to 1e-12 relative."""
import ctypes as C
import os

import numpy as np
import pytest

import gfship
from oracle import oracle as O
from poisson_cases import dirichlet_case, oracle_dirichlet_solve

pytestmark = pytest.mark.gpu
RTOL_SUM = 1e-12


def _pair(dim, level, side=None):
    od = O.Domain(dim, level, side)
    gd = gfship.Domain(dim, level, side)
    return od, gd


def _rand_fields(od, gd, names, rng, level=None):
    level = od.depth if level is None else level
    out = {}
    for nm in names:
        of, gf = od.field(), gd.variable()
        a = rng.standard_normal(of.level(level).shape)
        of.level(level)[...] = a
        gf.upload(a, level)
        out[nm] = (of, gf)
    return out


def _interior(a, dim):
    return a[(slice(1, -1),) * dim]


def _faces_equal(a, b, dim):
    """interior plus face ghosts (edge/corner ghosts are not part of the reference's data)."""
    if not np.array_equal(_interior(a, dim), _interior(b, dim)):
        return False
    for ax in range(dim):
        for s in (0, -1):
            sl = [slice(1, -1)] * dim
            sl[ax] = s
            if not np.array_equal(a[tuple(sl)], b[tuple(sl)]):
                return False
    return True


@pytest.mark.parametrize("dim,level", [(2, 3), (2, 6), (3, 2), (3, 4), (3, 5)])
def test_relax_sweeps_bit_exact(dim, level):
    L = O.lib()
    rng = np.random.default_rng(100 + 10 * dim + level)
    od, gd = _pair(dim, level)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia"], rng)
    f["dia"][0].leaf()[...] = 0.
    f["dia"][1].fill(0.)
    for omega in (1., 0.9):
        for _ in range(3):
            L.go_relax(od.ptr, dim, level, omega, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr)
            gd.relax(f["u"][1], f["rhs"][1], f["dia"][1], omega=omega)
        assert np.array_equal(f["u"][0].leaf(), f["u"][1].download())


@pytest.mark.parametrize("dim,level", [(2, 5), (3, 4)])
def test_relax_with_nonzero_dia_bit_exact(dim, level):
    L = O.lib()
    rng = np.random.default_rng(5)
    od, gd = _pair(dim, level)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia"], rng)
    a = np.abs(f["dia"][0].leaf()) + 0.1
    f["dia"][0].leaf()[...] = a
    f["dia"][1].upload(a)
    L.go_relax(od.ptr, dim, level, 1., f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr)
    gd.relax(f["u"][1], f["rhs"][1], f["dia"][1])
    assert np.array_equal(f["u"][0].leaf(), f["u"][1].download())


@pytest.mark.parametrize("dim,level", [(2, 6), (3, 5)])
def test_residual_and_norm(dim, level):
    L = O.lib()
    rng = np.random.default_rng(11)
    od, gd = _pair(dim, level)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia", "res"], rng)
    L.go_residual(od.ptr, dim, level, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                  f["res"][0].ptr)
    gd.residual(f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
    assert np.array_equal(_interior(f["res"][0].leaf(), dim), _interior(f["res"][1].download(), dim))
    on = L.go_norm_residual(od.ptr, 0.7, f["res"][0].ptr)
    gn = gd.norm_residual(f["res"][1], 0.7)
    assert gn.infty == on.infty            # max is order independent: exact
    for k in ("bias", "first", "second"):
        assert getattr(gn, k) == pytest.approx(getattr(on, k), rel=RTOL_SUM, abs=1e-13)
    on = L.go_norm_variable(od.ptr, f["u"][0].ptr)
    gn = gd.norm_variable(f["u"][1])
    assert gn.infty == on.infty
    for k in ("bias", "first", "second"):
        assert getattr(gn, k) == pytest.approx(getattr(on, k), rel=RTOL_SUM, abs=1e-15)


SIDES = {
    "dirichlet": ([O.SIDE_BOUNDARY] * 6, O.BC_DIRICHLET),
    "neumann": ([O.SIDE_BOUNDARY] * 6, O.BC_NEUMANN),
    "symmetry": ([O.SIDE_BOUNDARY] * 6, O.BC_SYMMETRY),
    "periodic": ([O.SIDE_PERIODIC] * 6, O.BC_SYMMETRY),
    "mixed": ([O.SIDE_PERIODIC, O.SIDE_PERIODIC, O.SIDE_BOUNDARY, O.SIDE_BOUNDARY,
               O.SIDE_BOUNDARY, O.SIDE_BOUNDARY], O.BC_NEUMANN),
}


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("kind", sorted(SIDES))
def test_bc_ghost_fill(dim, kind):
    L = O.lib()
    level = 4
    side, bck = SIDES[kind]
    rng = np.random.default_rng(3)
    od, gd = _pair(dim, level, side)
    n = 1 << level
    for component in (-1, 0, 1):
        of, gf = od.field(component), gd.variable(component)
        a = rng.standard_normal(of.leaf().shape)
        # only interior values are inputs; ghosts start at zero on both sides
        b = np.zeros_like(a)
        b[(slice(1, -1),) * dim] = a[(slice(1, -1),) * dim]
        of.leaf()[...] = b
        gf.upload(b)
        for d in range(2 * dim):
            val = rng.standard_normal(n ** (dim - 1))
            of.set_bc(d, bck, val)
            gf.set_bc(d, bck, val)
        L.go_bc(of.ptr, of.ptr, level)
        gd.bc(gf)
        assert _faces_equal(of.leaf(), gf.download(), dim)
        L.go_homogeneous_bc(of.ptr, of.ptr, level)
        gd.homogeneous_bc(gf, gf)
        assert _faces_equal(of.leaf(), gf.download(), dim)


@pytest.mark.parametrize("dim,level,kind", [(2, 5, "dirichlet"), (2, 5, "periodic"),
                                            (3, 4, "dirichlet"), (3, 4, "periodic"),
                                            (3, 5, "neumann"), (3, 5, "mixed")])
def test_poisson_cycle_bit_exact(dim, level, kind):
    """One full V-cycle (restriction, LDS relax loops on coarse levels, prolongation,
    hyperplane sweeps, correction, BC, residual) against the oracle."""
    L = O.lib()
    side, bck = SIDES[kind]
    rng = np.random.default_rng(17)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia", "res"], rng)
    f["dia"][0].leaf()[...] = 0.
    f["dia"][1].fill(0.)
    n = 1 << level
    for d in range(2 * dim):
        val = rng.standard_normal(n ** (dim - 1))
        f["u"][0].set_bc(d, bck, val)
        f["u"][1].set_bc(d, bck, val)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    L.go_residual(od.ptr, dim, level, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                  f["res"][0].ptr)
    gd.residual(f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
    op, gp = od.params(), gd.params()
    for p in (op, gp):
        p.depth = level
    for _ in range(2):
        L.go_poisson_cycle(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                           f["res"][0].ptr)
        gd.poisson_cycle(gp, f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)
        assert np.array_equal(_interior(f["res"][0].leaf(), dim),
                              _interior(f["res"][1].download(), dim))


def test_poisson_solve_reproduces_reference_golden_res7(golden_dir):
    """The reference's own golden curve (test/poisson/res-7.ref, 2-D level 8) through the device
    path: every printed digit of the max residual after N cycles."""
    rows = [l.split() for l in open(os.path.join(golden_dir, "reference", "poisson_res-7.ref"))]
    rhs, faces, _ = dirichlet_case(2, 8)
    for row in rows[::2]:
        cyc = int(row[0])
        gd = gfship.Domain(2, 8)
        P, div, res, dia = (gd.variable() for _ in range(4))
        a = np.zeros((258, 258))
        a[1:-1, 1:-1] = rhs
        div.upload(a)
        for d, fv in enumerate(faces):
            P.set_bc(d, gfship.BC_DIRICHLET, fv)
        gd.bc(P)
        gd.poisson_coefficients()
        par = gd.params()
        par.tolerance, par.nitermin, par.nitermax = 1e-30, cyc, cyc
        gd.poisson_solve(par, P, div, res, dia, 1.)
        assert "%.3e" % par.residual.infty == row[2]
        # and bit-exact against the oracle
        od, oP, ores, opar, _ = oracle_dirichlet_solve(2, 8, cyc)
        assert par.residual.infty == opar.residual.infty
        assert np.array_equal(oP.interior(), P.download()[1:-1, 1:-1])
        gd.destroy()


@pytest.mark.parametrize("level", [5, 6])
def test_poisson_solve_3d_dirichlet_bit_exact(level):
    """3-D analogue of test/poisson (SURVEY.md 8d config B at a size the oracle finishes in
    seconds): solve loop incl. niter, norms, stall logic."""
    rhs, faces, ex = dirichlet_case(3, level)
    n = 1 << level
    gd = gfship.Domain(3, level)
    P, div, res, dia = (gd.variable() for _ in range(4))
    a = np.zeros((n + 2,) * 3)
    a[1:-1, 1:-1, 1:-1] = rhs
    div.upload(a)
    for d, fv in enumerate(faces):
        P.set_bc(d, gfship.BC_DIRICHLET, fv)
    gd.bc(P)
    gd.poisson_coefficients()
    par = gd.params()
    par.tolerance, par.nitermin, par.nitermax = 1e-30, 4, 4
    gd.poisson_solve(par, P, div, res, dia, 1.)
    od, oP, ores, opar, _ = oracle_dirichlet_solve(3, level, 4)
    assert par.niter == opar.niter == 4
    assert par.residual.infty == opar.residual.infty
    assert par.residual_before.infty == opar.residual_before.infty
    assert np.array_equal(oP.interior(), P.download()[1:-1, 1:-1, 1:-1])
    assert np.array_equal(ores.interior(), res.download()[1:-1, 1:-1, 1:-1])


def test_redblack_mode_converges_to_same_solution():
    """The opt-in red-black smoother has different iterates but the same fixed point: converged
    to round-off, P agrees with the exact-mode P to 1e-12 relative L-inf."""
    level = 5
    rhs, faces, ex = dirichlet_case(3, level)
    n = 1 << level
    sols = []
    for mode in (gfship.RELAX_EXACT, gfship.RELAX_REDBLACK):
        gd = gfship.Domain(3, level)
        gd.set_relax_mode(mode)
        P, div, res, dia = (gd.variable() for _ in range(4))
        a = np.zeros((n + 2,) * 3)
        a[1:-1, 1:-1, 1:-1] = rhs
        div.upload(a)
        for d, fv in enumerate(faces):
            P.set_bc(d, gfship.BC_DIRICHLET, fv)
        gd.bc(P)
        gd.poisson_coefficients()
        par = gd.params()
        par.tolerance, par.nitermin, par.nitermax = 1e-30, 30, 30
        gd.poisson_solve(par, P, div, res, dia, 1.)
        sols.append(P.download()[1:-1, 1:-1, 1:-1])
    err = np.abs(sols[0] - sols[1]).max() / np.abs(sols[0]).max()
    assert err < 1e-12, err


# ---------------------------------------------------------------------------------------------
# pipelined skewed-tile sweep (relax_skew.hip): levels with n >= 32 in 3-D
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("level,kind", [(5, "periodic"), (6, "dirichlet"), (6, "periodic"),
                                        (6, "mixed"), (7, "neumann"), (7, "periodic")])
def test_skew_sweep_bit_exact_vs_oracle(level, kind):
    """multi-tile levels (64^3 = 4x4 tiles, 128^3 = 8x8 tiles): hand-offs between tiles, box-side
    ghost streams of every BC kind, non-zero dia, several sweeps with BC in between"""
    L = O.lib()
    dim = 3
    side, bck = SIDES[kind]
    rng = np.random.default_rng(1000 + level)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia"], rng)
    for use_dia in (False, True):
        if use_dia:
            a = np.abs(f["dia"][0].leaf()) + 0.5
            f["dia"][0].leaf()[...] = a
            f["dia"][1].upload(a)
        else:
            f["dia"][0].leaf()[...] = 0.
            f["dia"][1].fill(0.)
        for d in range(2 * dim):
            f["u"][0].set_bc(d, bck)
            f["u"][1].set_bc(d, bck)
        for _ in range(2):
            L.go_homogeneous_bc(f["u"][0].ptr, f["u"][0].ptr, level)
            gd.homogeneous_bc(f["u"][1], f["u"][1])
            L.go_relax(od.ptr, dim, level, 1., f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr)
            gd.relax(f["u"][1], f["rhs"][1], f["dia"][1])
            assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim), (use_dia,)


def test_skew_sweep_equals_hyperplane_sweep_256():
    """full size (256^3, 16x16 tiles): the pipelined sweep against the independent
    one-launch-per-hyperplane implementation, bit for bit, over 3 sweeps"""
    level, dim = 8, 3
    n = 1 << level
    rng = np.random.default_rng(8)
    u0 = rng.standard_normal((n + 2,) * 3)
    r0 = rng.standard_normal((n + 2,) * 3)
    out = []
    for mode in (gfship.RELAX_EXACT, gfship.RELAX_EXACT_HYPERPLANE):
        gd = gfship.Domain(dim, level, [gfship.SIDE_PERIODIC] * 6)
        gd.set_relax_mode(mode)
        gd.poisson_coefficients()
        u, rhs, dia = gd.variable(), gd.variable(), gd.variable()
        u.upload(u0)
        rhs.upload(r0)
        for _ in range(3):
            gd.homogeneous_bc(u, u)
            gd.relax(u, rhs, dia)
        out.append(u.download()[1:-1, 1:-1, 1:-1])
        gd.destroy()
    assert np.array_equal(out[0], out[1])


# ---------------------------------------------------------------------------------------------
# relax loop with its sweeps pipelined in one launch (relax_skew_loop.hip): periodic 3-D levels
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("level,use_dia", [(5, False), (6, False), (6, True), (7, False), (7, True)])
def test_fused_relax_loop_cycles_bit_exact_vs_oracle(level, use_dia):
    """V-cycles on a triply periodic box: the relax loops of the levels with n >= 32 run as one
    launch each (4 sweeps in flight behind each other, periodic images through the wrap
    hand-offs / snapshots, progress words).  Fields, the ghost layer left by the last BC
    application and the residual must equal the oracle bit for bit."""
    L = O.lib()
    dim = 3
    side, bck = SIDES["periodic"]
    rng = np.random.default_rng(4000 + level)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia", "res"], rng)
    if use_dia:
        for l in range(level + 1):
            a = np.abs(rng.standard_normal(f["dia"][0].level(l).shape)) + 0.5
            f["dia"][0].level(l)[...] = a
            f["dia"][1].upload(a, l)
    else:
        for l in range(level + 1):
            f["dia"][0].level(l)[...] = 0.
            f["dia"][1].fill(0., l)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    L.go_residual(od.ptr, dim, level, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                  f["res"][0].ptr)
    gd.residual(f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
    op, gp = od.params(), gd.params()
    for p in (op, gp):
        p.depth = level
    for _ in range(2):
        L.go_poisson_cycle(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                           f["res"][0].ptr)
        gd.poisson_cycle(gp, f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)
        assert np.array_equal(_interior(f["res"][0].leaf(), dim),
                              _interior(f["res"][1].download(), dim))


@pytest.mark.parametrize("nrelax", [2, 3, 4])
def test_fused_relax_loop_equals_per_sweep_launches_256(nrelax):
    """full size (256^3, 16 x 16 tiles): one V-cycle with the relax loops fused against the
    one-launch-per-sweep schedule, bit for bit (interior, ghost layer, residual)"""
    level, dim = 8, 3
    n = 1 << level
    rng = np.random.default_rng(80 + nrelax)
    u0 = rng.standard_normal((n + 2,) * 3)
    r0 = rng.standard_normal((n + 2,) * 3)
    out = []
    for mode in (gfship.RELAX_EXACT, gfship.RELAX_EXACT_PER_SWEEP):
        gd = gfship.Domain(dim, level, [gfship.SIDE_PERIODIC] * 6)
        gd.set_relax_mode(mode)
        gd.poisson_coefficients()
        u, rhs, dia, res = gd.variable(), gd.variable(), gd.variable(), gd.variable()
        u.upload(u0)
        rhs.upload(r0)
        gd.bc(u)
        gd.residual(u, rhs, dia, res)
        par = gd.params()
        par.depth = level
        par.nrelax = nrelax
        gd.poisson_cycle(par, u, rhs, dia, res)
        out.append((u.download(), res.download()[1:-1, 1:-1, 1:-1]))
        gd.destroy()
    assert _faces_equal(out[0][0], out[1][0], 3)
    assert np.array_equal(out[0][1], out[1][1])


def test_four_wave_sweep_kernel_still_bit_exact(monkeypatch):
    """GFSHIP_SKEW_OLD=1 selects the older four-wave sweep kernel (relax_skew_kernel), kept as an
    independent implementation of the same sweep: mixed sides, 64^3, against the oracle"""
    monkeypatch.setenv("GFSHIP_SKEW_OLD", "1")
    L = O.lib()
    level, dim = 6, 3
    side, bck = SIDES["mixed"]
    rng = np.random.default_rng(77)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia"], rng)
    f["dia"][0].leaf()[...] = 0.
    f["dia"][1].fill(0.)
    for d in range(2 * dim):
        f["u"][0].set_bc(d, bck)
        f["u"][1].set_bc(d, bck)
    for _ in range(2):
        L.go_homogeneous_bc(f["u"][0].ptr, f["u"][0].ptr, level)
        gd.homogeneous_bc(f["u"][1], f["u"][1])
        L.go_relax(od.ptr, dim, level, 1., f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr)
        gd.relax(f["u"][1], f["rhs"][1], f["dia"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)


@pytest.mark.parametrize("level,kind,component", [(5, "dirichlet", -1), (6, "neumann", -1),
                                                  (5, "symmetry", 1), (6, "mixed", -1),
                                                  (5, "symmetry", -1)])
def test_fused_relax_loop_non_periodic_sides_bit_exact(level, kind, component):
    """the relax loop in one launch on boxes with Dirichlet / Neumann / symmetry sides (ghost lines
    = +- the tile's own side lines of the previous sweep; the sign of a symmetry side depends on
    the component of the variable) and on mixed periodic / non-periodic boxes, with a non-zero
    dia: V-cycles against the oracle, interior, ghost layer and residual"""
    L = O.lib()
    dim = 3
    side, bck = SIDES[kind]
    rng = np.random.default_rng(5000 + level)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["rhs", "dia", "res"], rng)
    ou, gu = od.field(component), gd.variable(component)
    a = rng.standard_normal(ou.leaf().shape)
    ou.leaf()[...] = a
    gu.upload(a)
    f["u"] = (ou, gu)
    for l in range(level + 1):
        a = np.abs(rng.standard_normal(f["dia"][0].level(l).shape)) * 0.3
        f["dia"][0].level(l)[...] = a
        f["dia"][1].upload(a, l)
    n = 1 << level
    for d in range(2 * dim):
        val = rng.standard_normal(n * n)
        f["u"][0].set_bc(d, bck, val)
        f["u"][1].set_bc(d, bck, val)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    L.go_residual(od.ptr, dim, level, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                  f["res"][0].ptr)
    gd.residual(f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
    op, gp = od.params(), gd.params()
    for p in (op, gp):
        p.depth = level
    for _ in range(2):
        L.go_poisson_cycle(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                           f["res"][0].ptr)
        gd.poisson_cycle(gp, f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)
        assert np.array_equal(_interior(f["res"][0].leaf(), dim),
                              _interior(f["res"][1].download(), dim))


@pytest.mark.parametrize("level,kind", [(6, "periodic"), (5, "mixed"), (7, "periodic")])
def test_one_wave_per_tile_loop_kernel_bit_exact(monkeypatch, level, kind):
    """GFSHIP_WAVE_LOOP=1 selects the experimental relax_wave_loop_kernel (one compute wave per tile,
    neighbours through DPP / ds_bpermute instead of LDS + barrier) for the fused
    loops with dia == 0: V-cycles against the oracle, interior, ghost layer and residual"""
    monkeypatch.setenv("GFSHIP_WAVE_LOOP", "1")
    L = O.lib()
    dim = 3
    side, bck = SIDES[kind]
    rng = np.random.default_rng(6000 + level)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia", "res"], rng)
    for l in range(level + 1):
        f["dia"][0].level(l)[...] = 0.
        f["dia"][1].fill(0., l)
    if kind != "periodic":
        n = 1 << level
        for d in range(2 * dim):
            val = rng.standard_normal(n * n)
            f["u"][0].set_bc(d, bck, val)
            f["u"][1].set_bc(d, bck, val)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    L.go_residual(od.ptr, dim, level, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                  f["res"][0].ptr)
    gd.residual(f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
    op, gp = od.params(), gd.params()
    for p in (op, gp):
        p.depth = level
    for _ in range(2):
        L.go_poisson_cycle(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                           f["res"][0].ptr)
        gd.poisson_cycle(gp, f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)
        assert np.array_equal(_interior(f["res"][0].leaf(), dim),
                              _interior(f["res"][1].download(), dim))


# ---------------------------------------------------------------------------------------------
# K8: gfs_poisson_coefficients with a GfsFunction alpha (variable density), src/poisson.c:756-901
# ---------------------------------------------------------------------------------------------

def _alpha_pair(od, gd, rng, kind):
    """face values of alpha as oracle Fields and device Variables (same arrays, ghosts included:
    the ghost entry in front of the first cell is its - face)"""
    dim, level = od.dim, od.depth
    oa, ga = [], []
    for c in range(dim):
        of, gf = od.field(), gd.variable()
        a = 0.5 + rng.random(of.leaf().shape)
        if kind == "periodic":
            # a periodic alpha: the - face of the first cell is the + face of the last one
            ax = dim - 1 - c
            sl0, sln = [slice(None)] * dim, [slice(None)] * dim
            sl0[ax], sln[ax] = 0, -2
            a[tuple(sl0)] = a[tuple(sln)]
        of.leaf()[...] = a
        gf.upload(a)
        oa.append(of)
        ga.append(gf)
    return oa, ga


@pytest.mark.parametrize("dim,level,kind", [(2, 5, "dirichlet"), (2, 4, "periodic"), (3, 4, "dirichlet"),
                                            (3, 4, "periodic"), (3, 3, "mixed"),
                                            # levels of the pipelined weighted sweeps: the tile kernel with the
                                            # six weight rows streamed through LDS (3-D, 32^3 and more: the whole
                                            # loop in one launch on the periodic box, sweep by sweep otherwise),
                                            # one launch per sweep by rows in 2-D
                                            (3, 5, "periodic"), (3, 6, "dirichlet"), (3, 5, "mixed"),
                                            (3, 6, "periodic"), (2, 7, "periodic"), (2, 8, "dirichlet")])
def test_poisson_coefficients_with_alpha_bit_exact(dim, level, kind):
    """the face weights of every level (leaf faces from alpha, coarser cells by face_coeff_from_below),
    then sweeps, residual, V-cycles and a solve with them: everything equal to the oracle's bits"""
    L = O.lib()
    side, bck = SIDES[kind]
    rng = np.random.default_rng(77 + 10 * dim + level)
    od, gd = _pair(dim, level, side)
    oa, ga = _alpha_pair(od, gd, rng, kind)
    od.poisson_coefficients_alpha(oa)
    gd.poisson_coefficients_alpha(ga)
    inner = (slice(1, -1),) * dim
    for l in range(level + 1):
        for d in range(2 * dim):
            assert np.array_equal(od.weight(d, l)[inner], gd.poisson_weight(d, l)[inner]), (l, d)
    assert not np.array_equal(od.weight(0, level)[inner], np.ones_like(od.weight(0, level)[inner]))
    f = _rand_fields(od, gd, ["u", "rhs", "dia", "res"], rng)
    f["dia"][0].leaf()[...] = 0.
    f["dia"][1].fill(0.)
    n = 1 << level
    for d in range(2 * dim):
        val = rng.standard_normal(n ** (dim - 1))
        f["u"][0].set_bc(d, bck, val)
        f["u"][1].set_bc(d, bck, val)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    for _ in range(2):
        L.go_relax(od.ptr, dim, level, 1., f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr)
        gd.relax(f["u"][1], f["rhs"][1], f["dia"][1])
        assert np.array_equal(f["u"][0].leaf()[inner], f["u"][1].download()[inner])
    L.go_residual(od.ptr, dim, level, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                  f["res"][0].ptr)
    gd.residual(f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
    assert np.array_equal(f["res"][0].leaf()[inner], f["res"][1].download()[inner])
    op, gp = od.params(), gd.params()
    for p in (op, gp):
        p.depth = level
    for _ in range(2):
        L.go_poisson_cycle(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                           f["res"][0].ptr)
        gd.poisson_cycle(gp, f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)
        assert np.array_equal(f["res"][0].leaf()[inner], f["res"][1].download()[inner])
    op.tolerance = gp.tolerance = 1e-30
    op.nitermin = gp.nitermin = op.nitermax = gp.nitermax = 3
    L.go_poisson_solve(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["res"][0].ptr,
                       f["dia"][0].ptr, 0.7)
    gd.poisson_solve(gp, f["u"][1], f["rhs"][1], f["res"][1], f["dia"][1], 0.7)
    assert gp.niter == op.niter == 3
    assert gp.residual.infty == op.residual.infty
    assert np.array_equal(f["u"][0].leaf()[inner], f["u"][1].download()[inner])
    # back to alpha = NULL: the unit-weight kernels again
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    L.go_relax(od.ptr, dim, level, 1., f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr)
    gd.relax(f["u"][1], f["rhs"][1], f["dia"][1])
    assert np.array_equal(f["u"][0].leaf()[inner], f["u"][1].download()[inner])


# ---------------------------------------------------------------------------------------------
# the three implementations of the pipelined sweep (one line per thread; 2 x 2 lines per lane with
# the rows streamed through registers, or through an LDS ring by a fourth wave) on every size and
# kind of side, whatever the default choice per level is
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("impl", ["lines", "patch_regs", "patch_ring"])
@pytest.mark.parametrize("level,kind,use_dia", [(5, "periodic", False), (6, "dirichlet", False),
                                                (6, "mixed", True), (6, "periodic", True),
                                                (7, "neumann", False)])
def test_every_sweep_implementation_bit_exact(impl, level, kind, use_dia, monkeypatch):
    monkeypatch.delenv("GFSHIP_SKEW_LINES", raising=False)
    monkeypatch.delenv("GFSHIP_PATCH_REGS", raising=False)
    monkeypatch.setenv("GFSHIP_PATCH_MIN_N", "32")
    if impl == "lines":
        monkeypatch.setenv("GFSHIP_SKEW_LINES", "1")
    elif impl == "patch_regs":
        monkeypatch.setenv("GFSHIP_PATCH_REGS", "1")
    L = O.lib()
    dim = 3
    side, bck = SIDES[kind]
    rng = np.random.default_rng(9000 + level)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia", "res"], rng)
    for l in range(level + 1):
        if use_dia:
            a = np.abs(rng.standard_normal(f["dia"][0].level(l).shape)) + 0.5
            f["dia"][0].level(l)[...] = a
            f["dia"][1].upload(a, l)
        else:
            f["dia"][0].level(l)[...] = 0.
            f["dia"][1].fill(0., l)
    n = 1 << level
    for d in range(2 * dim):
        val = rng.standard_normal(n * n)
        f["u"][0].set_bc(d, bck, val)
        f["u"][1].set_bc(d, bck, val)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    # single sweeps with the BC kernel around them (mirror mode) ...
    for _ in range(2):
        L.go_homogeneous_bc(f["u"][0].ptr, f["u"][0].ptr, level)
        gd.homogeneous_bc(f["u"][1], f["u"][1])
        L.go_relax(od.ptr, dim, level, 1., f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr)
        gd.relax(f["u"][1], f["rhs"][1], f["dia"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)
    # ... and V-cycles (fused loops where the sides allow it)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    L.go_residual(od.ptr, dim, level, f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                  f["res"][0].ptr)
    gd.residual(f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
    op, gp = od.params(), gd.params()
    for p in (op, gp):
        p.depth = level
    for _ in range(2):
        L.go_poisson_cycle(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["dia"][0].ptr,
                           f["res"][0].ptr)
        gd.poisson_cycle(gp, f["u"][1], f["rhs"][1], f["dia"][1], f["res"][1])
        assert _faces_equal(f["u"][0].leaf(), f["u"][1].download(), dim)
        assert np.array_equal(_interior(f["res"][0].leaf(), dim),
                              _interior(f["res"][1].download(), dim))
    gd.destroy()


def test_a_hand_off_that_never_arrives_is_reported_and_the_domain_keeps_working(monkeypatch):
    """include/gfship.h: the tiles of a fused relax loop wait for each other's lines with bounded waits.
    A hand-off that never comes (here: tile 0 of the 128^3 level publishes nothing in its first sweep,
    GFSHIP_FAULT_DROP_HANDOFF) becomes an error of the solve instead of a hang; the error is not
    sticky: the same domain then solves the same problem (with one launch per sweep) bit for bit."""
    L = O.lib()
    dim, level = 3, 7
    side, bck = SIDES["periodic"]
    rng = np.random.default_rng(99)
    od, gd = _pair(dim, level, side)
    L.go_poisson_coefficients(od.ptr)
    gd.poisson_coefficients()
    f = _rand_fields(od, gd, ["u", "rhs", "dia", "res"], rng)
    for l in range(level + 1):
        f["dia"][0].level(l)[...] = 0.
        f["dia"][1].fill(0., l)
    u0 = f["u"][0].leaf().copy()
    op, gp = od.params(), gd.params()
    for p in (op, gp):
        p.tolerance, p.nitermin, p.nitermax = 1e-30, 1, 1
    gd.bc(f["u"][1])
    gd.poisson_solve(gp, f["u"][1], f["rhs"][1], f["res"][1], f["dia"][1], 1.)     # warm: trial runs done
    monkeypatch.setenv("GFSHIP_FAULT_DROP_HANDOFF", "0")
    f["u"][1].upload(u0)
    gd.bc(f["u"][1])
    with pytest.raises(gfship.GfshipError, match="hand-off"):
        gd.poisson_solve(gp, f["u"][1], f["rhs"][1], f["res"][1], f["dia"][1], 1.)
    monkeypatch.delenv("GFSHIP_FAULT_DROP_HANDOFF")
    # the same domain, the same data: equal to the oracle
    f["u"][1].upload(u0)
    L.go_bc(f["u"][0].ptr, f["u"][0].ptr, level)
    gd.bc(f["u"][1])
    L.go_poisson_solve(od.ptr, C.byref(op), f["u"][0].ptr, f["rhs"][0].ptr, f["res"][0].ptr, f["dia"][0].ptr, 1.)
    gd.poisson_solve(gp, f["u"][1], f["rhs"][1], f["res"][1], f["dia"][1], 1.)
    assert np.array_equal(_interior(f["u"][0].leaf(), dim), _interior(f["u"][1].download(), dim))
    assert gp.residual.infty == op.residual.infty
