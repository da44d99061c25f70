"""Pins the oracle's Poisson multigrid on the reference's own golden files
(test/poisson/res-7.ref, error.ref; checked by test/poisson/poisson.sh:103-112)."""
import os

import numpy as np
import pytest

from poisson_cases import oracle_dirichlet_solve, error_norm_unbiased


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, "reference", name)) as f:
        return [line.split() for line in f if line.strip()]


def test_residual_curve_matches_res7_ref(golden_dir):
    """res-7.ref column 3: max residual after N = 0..10 V-cycles at LEVEL=8 (poisson.sh:15).
    The reference prints '% 10.3e': match every printed digit."""
    rows = _load(golden_dir, "poisson_res-7.ref")
    for row in rows:
        cyc, ref = int(row[0]), row[2]
        _, _, _, par, _ = oracle_dirichlet_solve(2, 8, cyc)
        assert "%.3e" % par.residual.infty == ref, (cyc, par.residual.infty, ref)
        assert par.niter == cyc


def test_error_norms_match_error_ref(golden_dir):
    """error.ref: L1, L2, Linf error after 10 cycles for levels 3..8."""
    for row in _load(golden_dir, "poisson_error.ref"):
        level = int(row[0])
        dom, P, _, _, ex = oracle_dirichlet_solve(2, level, 10)
        nm = error_norm_unbiased(dom, P, ex)
        got = ["%.3e" % v for v in (nm.first, nm.second, nm.infty)]
        assert got == row[1:4], (level, got, row)


def test_order_of_convergence_matches_order_ref(golden_dir):
    """order.ref: the convergence orders poisson.sh:32-39 derives (awk) from the printed rows of the
    `error' file -- here from the oracle's own errors, printed as the reference prints them."""
    import math
    rows = []
    for row in _load(golden_dir, "poisson_error.ref"):
        level = int(row[0])
        dom, P, _, _, ex = oracle_dirichlet_solve(2, level, 10)
        nm = error_norm_unbiased(dom, P, ex)
        rows.append([level] + [float("%.3e" % v) for v in (nm.first, nm.second, nm.infty)])
    ref = _load(golden_dir, "poisson_order.ref")
    assert len(ref) == len(rows) - 1
    for i in range(1, len(rows)):
        got = ["%d" % rows[i][0]] + ["%.6g" % (math.log(rows[i - 1][c] / rows[i][c]) / math.log(2.))
                                     for c in (1, 2, 3)]
        assert got == ref[i - 1], (got, ref[i - 1])


def test_lexicographic_sweep_is_bit_identical_to_tree_order():
    """The device kernels sweep hyperplanes of the (+x,-y,-z) lexicographic order; this must be
    bit-identical to the reference's tree pre-order sweep (hard part 1)."""
    from oracle import oracle as O
    L = O.lib()
    rng = np.random.default_rng(7)
    for dim, level in ((2, 5), (3, 4)):
        dom = O.Domain(dim, level)
        L.go_poisson_coefficients(dom.ptr)
        u1, u2, rhs, dia = dom.field(), dom.field(), dom.field(), dom.field()
        a = rng.standard_normal(u1.leaf().shape)
        u1.leaf()[...] = a
        u2.leaf()[...] = a
        rhs.leaf()[...] = rng.standard_normal(a.shape)
        for _ in range(3):
            L.go_relax(dom.ptr, dim, level, 1., u1.ptr, rhs.ptr, dia.ptr)
            L.go_relax_lexicographic(dom.ptr, dim, level, 1., u2.ptr, rhs.ptr, dia.ptr)
            assert np.array_equal(u1.leaf(), u2.leaf())
