"""Seeded synthetic inputs in the reference's host layouts (host-side utility; numpy only).

The gauge field is defined per *global* time-slice, so every rank of a T-split lattice can build
its own slab -- including the two halo slices the reference obtains through xchange_gauge
(benchmark.c:250-253) -- without communication, and a single-rank run of the same global
lattice sees identical links.  Link construction follows random_su3 (start.c:387-425):
two random vectors, Gram-Schmidt, third row = conj(cross product).
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def _threads():
    """Worker threads for the per-time-slice generators (numpy releases the GIL inside its kernels): the CPUs this process may use,
    at most 16 -- a 64 x 32^3 gauge field took 47 s on one thread, most of a multi-rank bench run."""
    e = os.environ.get("TMLQCD_SYNTH_THREADS")      # the ranks of a multi-rank bench run share the node's CPUs: bench.py sets CPUs / ranks
    if e and e.isdigit() and int(e) > 0:
        return min(16, int(e))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def _su3(rng, n):
    z1 = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    z2 = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    z1 /= np.linalg.norm(z1, axis=1, keepdims=True)
    z2 -= (np.conj(z1) * z2).sum(axis=1, keepdims=True) * z1
    z2 /= np.linalg.norm(z2, axis=1, keepdims=True)
    u = np.stack([z1, z2, np.conj(np.cross(z1, z2))], axis=1)
    out = np.empty((n, 3, 3, 2), dtype=np.float64)
    out[..., 0], out[..., 1] = u.real, u.imag
    return out


def gauge_slice(seed, t_global, LX, LY, LZ):
    """Links of one global time-slice: [LX*LY*LZ][4][3][3][2]."""
    rng = np.random.default_rng([seed, 1, t_global])
    return _su3(rng, LX * LY * LZ * 4).reshape(LX * LY * LZ, 4, 3, 3, 2)


def gauge_field(seed, T, LX, LY, LZ, nproc_t=1, proc_t=0):
    """g_gauge_field of one rank: [VOLUMEPLUSRAND][4][3][3][2], lexicographic (geometry_eo.c:290),
    with the t = T and t = -1 halo slabs appended when nproc_t > 1 (geometry_eo.c:292-299)."""
    Tg, XYZ = T * nproc_t, LX * LY * LZ
    ts = [proc_t * T + t for t in range(T)]
    if nproc_t > 1:
        ts += [(proc_t * T + T) % Tg, (proc_t * T - 1) % Tg]
    out = np.empty((len(ts) * XYZ, 4, 3, 3, 2), dtype=np.float64)

    def fill(j):
        out[j * XYZ:(j + 1) * XYZ] = gauge_slice(seed, ts[j], LX, LY, LZ)
    with ThreadPoolExecutor(_threads()) as ex:      # every slice has its own seeded stream: the field does not depend on the thread count
        list(ex.map(fill, range(len(ts))))
    return out


def spinor_slice(seed, t_global, LX, LY, LZ):
    rng = np.random.default_rng([seed, 2, t_global])
    return rng.standard_normal((LX * LY * LZ, 4, 3, 2))


def spinor_field_eo(seed, parity, T, LX, LY, LZ, nproc_t=1, proc_t=0):
    """Gaussian spinor on the sites of one parity, e/o order (geometry_eo.c:869-885): [V/2][4][3][2]."""
    out = []
    z, y, x = np.meshgrid(np.arange(LZ), np.arange(LY), np.arange(LX), indexing="ij")
    sxyz = (x + y + z).transpose(2, 1, 0).reshape(-1)  # lexicographic x,y,z with z fastest
    for t in range(T):
        tg = proc_t * T + t
        full = spinor_slice(seed, tg, LX, LY, LZ)
        out.append(full[((sxyz + tg) & 1) == parity])
    return np.ascontiguousarray(np.concatenate(out, axis=0))


def eo2lexic_even(T, LX, LY, LZ):
    """Lexicographic indices of the even sites in e/o order (geometry_eo.c:869-885), single rank."""
    t, x, y, z = np.meshgrid(np.arange(T), np.arange(LX), np.arange(LY), np.arange(LZ), indexing="ij")
    par = ((t + x + y + z) & 1).reshape(-1)
    return np.nonzero(par == 0)[0]


def clover_blocks(seed, T, LX, LY, LZ, mu, scale=0.15):
    """Synthetic clover arrays in the reference's layouts (host-side input like the gauge field):
    sw[V][3][2][3][3][2]     hermitian 6x6 blocks 1 + T per site and chirality (operator/clover_term.c:60-87)
    sw_inv[V][4][2][3][3][2] (1 + T +- i mu g5)^-1 on the even sites, +mu set in [0,V/2), -mu set in [V/2,V)
                             (operator/clover_invert.c:164-257)."""
    rng = np.random.default_rng([seed, 3])
    V = T * LX * LY * LZ
    Vh = V // 2
    H = rng.standard_normal((V, 2, 6, 6)) + 1j * rng.standard_normal((V, 2, 6, 6))
    H = np.eye(6) + scale * 0.5 * (H + np.conj(np.swapaxes(H, -1, -2)))
    sw = np.zeros((V, 3, 2, 3, 3), dtype=np.complex128)
    sw[:, 0], sw[:, 1], sw[:, 2] = H[:, :, 0:3, 0:3], H[:, :, 0:3, 3:6], H[:, :, 3:6, 3:6]
    swi = np.zeros((V, 4, 2, 3, 3), dtype=np.complex128)
    ev = eo2lexic_even(T, LX, LY, LZ)
    for s, sgn in ((0, +1.0), (1, -1.0)):
        for chi, g5 in ((0, +1.0), (1, -1.0)):
            Mi = np.linalg.inv(H[ev, chi] + 1j * sgn * g5 * mu * np.eye(6))
            sl = slice(s * Vh, (s + 1) * Vh)
            swi[sl, 0, chi], swi[sl, 1, chi] = Mi[:, 0:3, 0:3], Mi[:, 0:3, 3:6]
            swi[sl, 2, chi], swi[sl, 3, chi] = Mi[:, 3:6, 3:6], Mi[:, 3:6, 0:3]

    def pack(a):
        out = np.empty(a.shape + (2,), dtype=np.float64)
        out[..., 0], out[..., 1] = a.real, a.imag
        return out
    return pack(sw), pack(swi)
