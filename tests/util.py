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
    """Synthetic clover blocks in the reference's layouts: sw[V][3][2] = hermitian 6x6 blocks 1 + T (T small, random)
    per site and chirality, sw_inv[V][4][2] = (1 + T +- i mu g5)^-1 on the EVEN sites, +mu set first, -mu set at V/2
    (operator/clover_invert.c:164-257).  `orc` supplies eo2lexic."""
    rng = np.random.default_rng(seed)
    V, Vh = orc.V, orc.Vh
    H = rng.standard_normal((V, 2, 6, 6)) + 1j * rng.standard_normal((V, 2, 6, 6))
    H = np.eye(6) + scale * 0.5 * (H + np.conj(np.swapaxes(H, -1, -2)))
    sw = np.zeros((V, 3, 2, 3, 3), dtype=np.complex128)
    sw[:, 0] = H[:, :, 0:3, 0:3]
    sw[:, 1] = H[:, :, 0:3, 3:6]
    sw[:, 2] = H[:, :, 3:6, 3:6]
    swi = np.zeros((V, 4, 2, 3, 3), dtype=np.complex128)
    ev = orc.eo2lexic()[:Vh]
    for s, sgn in ((0, +1.0), (1, -1.0)):
        for chi, g5 in ((0, +1.0), (1, -1.0)):
            M = H[ev, chi] + 1j * sgn * g5 * mu * np.eye(6)
            Mi = np.linalg.inv(M)
            sl = slice(s * Vh, (s + 1) * Vh)
            swi[sl, 0, chi] = Mi[:, 0:3, 0:3]
            swi[sl, 1, chi] = Mi[:, 0:3, 3:6]
            swi[sl, 2, chi] = Mi[:, 3:6, 3:6]
            swi[sl, 3, chi] = Mi[:, 3:6, 0:3]

    def pack(a):
        out = np.empty(a.shape + (2,), dtype=np.float64)
        out[..., 0], out[..., 1] = a.real, a.imag
        return np.ascontiguousarray(out)
    return pack(sw), pack(swi)
