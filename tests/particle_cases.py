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
