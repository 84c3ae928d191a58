"""GPU: the drop-in's residency modes under a seeded random program -- device calls (stencil, operator compositions, linalg on whole
fields, on site prefixes, on lexicographic pairs), host loads of a few sites / a stride / everything, host stores into a few sites /
everything, host threads summing a field -- run once in COHERENT mode (every call uploads and downloads: the plain drop-in
semantics) and once in LAZY mode (the same calls, data kept in HBM, the library told by page faults alone).  The device kernels are
the same in both modes, so every number the host sees and every array at the end must agree BIT FOR BIT: what differs is only when
data moves, and that is exactly what this guards (mirror shapes replacing one another at the same address, shared edge pages, pages
fetched one by one and then whole, stores into write-protected inputs)."""
import ctypes as C

import numpy as np
import pytest

from tests.test_gpu_lazy import COHERENT, LAZY, VP, _p, prog  # noqa: F401  (prog: the fixture -- fields back to back in one block)
from tests.util import random_spinor

pytestmark = pytest.mark.gpu


def _program(stub, d, f, block, V, N, seed, nops):
    rng = np.random.default_rng(seed)
    nf = len(f)
    pairs = [np.frombuffer(block, dtype=np.float64, count=V * 24, offset=f[2 * j].ctypes.data - block.ctypes.data).reshape(V, 4, 3, 2) for j in range(nf // 2)]
    for i in range(nf):
        f[i][:] = random_spinor(500 + i, N)
    seen = []
    for step in range(nops):
        op = int(rng.integers(0, 14))
        a, b = (int(x) for x in rng.choice(nf, size=2, replace=False))
        ieo = int(rng.integers(0, 2))
        c = float(rng.uniform(-0.8, 0.8))
        if op == 0:
            d.Hopping_Matrix(ieo, _p(f[a]), _p(f[b]))
        elif op == 1:
            d.Qtm_pm_psi(_p(f[a]), _p(f[b]))
        elif op == 2:
            d.assign_add_mul_r(_p(f[a]), _p(f[b]), c, N)
        elif op == 3:
            seen.append(d.square_norm(_p(f[a]), N, 1))
        elif op == 4:                                          # the host reads a few sites
            s0 = int(rng.integers(0, N - 40))
            seen.append(float(f[a][s0:s0 + int(rng.integers(1, 40))].sum()))
        elif op == 5:                                          # ... a stride through the whole field (pages one by one, then the rest)
            seen.append(float(f[a][::int(rng.integers(200, 2000))].sum()))
        elif op == 6:                                          # the host stores into a few sites of a field (input or stale output alike)
            s0 = int(rng.integers(0, N - 10))
            f[a][s0:s0 + 5] *= 1.0 + c
        elif op == 7:                                          # the host overwrites a field completely
            f[a][:] = random_spinor(2000 + step, N)
        elif op == 8:                                          # linalg on a site prefix (block volumes, test_linalg_spinor.c's N = 1000)
            n = int(rng.integers(1, N))
            d.assign_add_mul_r(_p(f[a]), _p(f[b]), c, n)
        elif op == 9:                                          # a lexicographic pair as ONE full field (and a prefix longer than VOLUME/2)
            ja, jb = (int(x) for x in rng.choice(nf // 2, size=2, replace=False))
            if rng.random() < 0.5:
                d.D_psi(_p(pairs[ja]), _p(pairs[jb]))
            else:
                d.assign_add_mul_r(_p(pairs[ja]), _p(pairs[jb]), c, N + int(rng.integers(1, N)))
        elif op == 10:
            seen.append(d.scalar_prod_r(_p(f[a]), _p(f[b]), N, 1))
        elif op == 11:                                         # host threads read a (possibly stale) field at the same time
            seen.append(stub.stub_host_sum_threads(_p(f[a]), N, 6))
        elif op == 12:
            d.mul_r(_p(f[a]), 0.5 + abs(c), _p(f[b]), N)
        else:
            d.gamma5(_p(f[a]), _p(f[b]), N)
        if step % 9 == 8:                                      # keep the numbers O(1) (host side: loads and stores of whole fields)
            for g in f:
                n = float(np.sqrt((g * g).sum() / N))
                if n > 0:
                    g *= 1.0 / n
    return seen, [g.copy() for g in f]


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_lazy_mode_agrees_with_coherent_mode_bit_for_bit(prog, seed):
    stub, d, orc, f, (T, L, V, N), block = prog
    f = f[:4]                                                  # two lexicographic pairs
    d.D_psi.argtypes = [VP, VP]
    d.assign_add_mul_r.argtypes = [VP, VP, C.c_double, C.c_int]
    d.scalar_prod_r.restype = C.c_double; d.scalar_prod_r.argtypes = [VP, VP, C.c_int, C.c_int]
    d.mul_r.argtypes = [VP, C.c_double, VP, C.c_int]
    d.gamma5.argtypes = [VP, VP, C.c_int]
    stub.stub_host_sum_threads.restype = C.c_double; stub.stub_host_sum_threads.argtypes = [VP, C.c_int, C.c_int]
    out = {}
    for mode in (COHERENT, LAZY):
        d.tmlqcd_hip_set_residency(mode)
        out[mode] = _program(stub, d, f, block, V, N, seed, 250)
        d.tmlqcd_hip_set_residency(COHERENT)                   # leaving lazy mode: every array current and unwatched
    (s0, a0), (s1, a1) = out[COHERENT], out[LAZY]
    assert len(s0) == len(s1)
    for k, (x, y) in enumerate(zip(s0, s1)):
        assert x == y, (seed, k, x, y)
    for i, (x, y) in enumerate(zip(a0, a1)):
        assert np.array_equal(x, y), (seed, i, float(np.abs(x - y).max()))
    st = (C.c_ulong * 4)(); d.tmlqcd_hip_lazy_stats(st)
    assert st[0] > 0 and st[3] > 0                             # the lazy run did take faults, stores among them
