"""Deterministic tracer seeds shared by oracle and GPU tests (positions
uniform in [-1/2, 1/2)^3 from a fixed-seed 64-bit LCG, seed 12345, ids 1..Np)."""
import numpy as np


def lcg_positions(npart, dim=3, seed=12345):
    # Knuth MMIX 64-bit LCG; the top 53 bits give a double in [0, 1)
    a, c, m = 6364136223846793005, 1442695040888963407, 1 << 64
    x = seed
    out = np.empty((npart, 3))
    for q in range(npart):
        for d in range(3):
            x = (a * x + c) % m
            out[q, d] = (x >> 11) / float(1 << 53) - 0.5
    if dim == 2:
        out[:, 2] = 0.
    return out, np.arange(1, npart + 1, dtype=np.uint32)


def lcg_positions_fast(npart, dim=3, seed=12345):
    """the same sequence as lcg_positions, vectorised: x_k = a^k x_0 + c (1 + a + ... + a^(k-1))
    in uint64 arithmetic (wraps mod 2^64)"""
    m = 3 * npart
    with np.errstate(over="ignore"):
        a = np.full(m, 6364136223846793005, dtype=np.uint64)
        a[0] = 1
        pw = np.cumprod(a)                      # a^0 .. a^(m-1)
        s = np.cumsum(pw)                       # 1 + a + ... + a^(k-1), k = 1..m
        ak = pw * np.uint64(6364136223846793005)  # a^1 .. a^m
        x = ak * np.uint64(seed) + np.uint64(1442695040888963407) * s
    out = ((x >> np.uint64(11)).astype(np.float64) / float(1 << 53) - 0.5).reshape(npart, 3)
    if dim == 2:
        out[:, 2] = 0.
    return out, np.arange(1, npart + 1, dtype=np.uint32)
