"""Shared helpers for the tests: seeded synthetic inputs in the reference's AoS layouts."""
import numpy as np


def random_su3(rng, n):
    """n random SU(3) matrices, float64 [n][3][3][2] (row-major c00,c01,.. as su3.h:40-43).

    Same construction as the reference's random_su3 (start.c:387-425): two random unit
    vectors, Gram-Schmidt, third row = conj(cross product)."""
    z1 = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    z2 = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    z1 /= np.linalg.norm(z1, axis=1, keepdims=True)
    z2 -= (np.conj(z1) * z2).sum(axis=1, keepdims=True) * z1
    z2 /= np.linalg.norm(z2, axis=1, keepdims=True)
    z3 = np.conj(np.cross(z1, z2))
    u = np.stack([z1, z2, z3], axis=1)
    out = np.empty((n, 3, 3, 2), dtype=np.float64)
    out[..., 0] = u.real
    out[..., 1] = u.imag
    return out


def random_gauge(seed, VPR):
    rng = np.random.default_rng(seed)
    return random_su3(rng, VPR * 4).reshape(VPR, 4, 3, 3, 2)


def random_spinor(seed, n):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, 4, 3, 2))


def rel_err(a, b):
    """max |a-b| / max |b|  -- the per-site tolerance measure of BASELINE.md §3.4."""
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


TOL = 1e-13  # fp64 tolerance stated in BASELINE.md §3.4 / SURVEY.md §8c


def random_clover(seed, orc, mu, scale=0.15):
    """Synthetic clover blocks (tmlqcd_amd.synthetic.clover_blocks) for the lattice of oracle `orc`."""
    from tmlqcd_amd import synthetic as syn
    assert np.array_equal(syn.eo2lexic_even(orc.T, orc.LX, orc.LY, orc.LZ), orc.eo2lexic()[:orc.Vh])
    return syn.clover_blocks(seed, orc.T, orc.LX, orc.LY, orc.LZ, mu, scale)
