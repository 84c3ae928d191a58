"""GPU: the drop-in's LAZY residency mode -- an UNMODIFIED host program keeps its fields in HBM, the library learns from page
faults when the host reads or writes a mirrored array (dropin.cpp, "lazy coherence").  Host arrays are laid out as
init_spinor_field does (one block, fields back to back, base 32-byte aligned only), so every field shares its first and last
page with its neighbours."""
import ctypes as C
import time

import numpy as np
import pytest

from tests.util import TOL, random_gauge, random_spinor, rel_err

pytestmark = pytest.mark.gpu
VP = C.c_void_p
LAZY, RESIDENT, COHERENT = 2, 1, 0


def _p(a):
    return a.ctypes.data_as(VP)


@pytest.fixture()
def prog(host_stub):
    from oracle.oraclebind import Oracle
    stub, d = host_stub
    T, L = 16, 16
    kappa, mu, theta = 0.129, 0.013, (1.0, 0.0, 0.0, 0.0)
    V = T * L ** 3
    gptr = stub.stub_init(T, L, L, L)
    g = random_gauge(61, V)
    C.memmove(gptr, _p(g), g.nbytes)
    stub.stub_boundary(kappa, *theta)
    stub.stub_set_mu(mu)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=8)
    orc.set_gauge(g)
    d.Hopping_Matrix.argtypes = [C.c_int, VP, VP]
    d.Qtm_pm_psi.argtypes = [VP, VP]
    d.square_norm.restype = C.c_double; d.square_norm.argtypes = [VP, C.c_int, C.c_int]
    d.cg_her.restype = C.c_int; d.cg_her.argtypes = [VP, VP, C.c_int, C.c_double, C.c_int, C.c_int, VP]
    d.tmlqcd_hip_set_residency.argtypes = [C.c_int]
    d.tmlqcd_hip_sync_to_host.argtypes = [VP]
    stub.stub_benchmark_loop.restype = C.c_double
    stub.stub_benchmark_loop.argtypes = [VP, VP, VP, VP, C.c_int]
    stub.stub_host_scale.argtypes = [VP, C.c_int, C.c_double]
    stub.stub_host_sample.restype = C.c_double; stub.stub_host_sample.argtypes = [VP, C.c_int, C.c_int]
    # init_spinor_field.c:48: one calloc for all fields, base aligned to ALIGN_BASE (32 bytes) only
    N = V // 2
    nf = 5
    block = np.zeros(nf * N * 192 + 4096 + 64, dtype=np.uint8)
    base = (block.ctypes.data + 31) // 32 * 32 + 32           # 32-byte aligned, NOT page aligned
    off = base - block.ctypes.data
    fields = [np.frombuffer(block, dtype=np.float64, count=N * 24, offset=off + i * N * 192).reshape(N, 4, 3, 2) for i in range(nf)]
    assert fields[0].ctypes.data % 4096 != 0
    yield stub, d, orc, fields, (T, L, V, N), block
    d.tmlqcd_hip_set_residency(COHERENT)
    d.tmlqcd_hip_finalize()


def test_unmodified_benchmark_loop_runs_resident_and_stays_correct(prog):
    stub, d, orc, f, (T, L, V, N), _ = prog
    src = random_spinor(71, N)
    hop = C.cast(d.Hopping_Matrix, VP)
    iters = 30
    # reference result: the loop is idempotent (f1 = H f0, f2 = H f1 every iteration)
    k = orc.new_field(); k[:N] = src
    r1, r2 = orc.new_field(), orc.new_field()
    orc.Hopping_Matrix(0, r1, k); orc.Hopping_Matrix(1, r2, r1)
    want_sum = iters * r2[0, 0, 0, 0]
    times = {}
    for mode in (LAZY, COHERENT):          # (lazy first: arrays the runtime has once copied from / to directly stay registered with the driver)
        d.tmlqcd_hip_set_residency(mode)
        f[0][:] = src; f[1][:] = 0; f[2][:] = 0
        stub.stub_benchmark_loop(hop, _p(f[0]), _p(f[1]), _p(f[2]), 2)
        t0 = time.perf_counter()
        got = stub.stub_benchmark_loop(hop, _p(f[0]), _p(f[1]), _p(f[2]), iters)
        times[mode] = time.perf_counter() - t0
        st = (C.c_ulong * 4)(); d.tmlqcd_hip_lazy_stats(st); print("mode", mode, "seconds", times[mode], "lazy stats", list(st))
        assert abs(got - want_sum) <= 1e-12 * abs(want_sum) + 1e-12
        # the host reads the outputs afterwards (numpy loads fault page by page, then the rest in one go)
        assert rel_err(f[2], r2[:N]) < TOL and rel_err(f[1], r1[:N]) < TOL
        assert np.array_equal(f[0], src)
    assert times[LAZY] < 0.5 * times[COHERENT], times       # 16^4: PCIe both ways per call against one page per iteration


def test_host_stores_and_partial_reads_between_device_calls(prog):
    stub, d, orc, f, (T, L, V, N), _ = prog
    d.tmlqcd_hip_set_residency(LAZY)
    src = random_spinor(72, N)
    f[0][:] = src
    d.Hopping_Matrix(0, _p(f[1]), _p(f[0]))                  # f0 uploaded (now write-protected), f1 stale on the host
    k = orc.new_field(); k[:N] = src
    r1 = orc.new_field(); orc.Hopping_Matrix(0, r1, k)
    # a strided host read of the stale output: the first pages one by one, then the whole field
    assert abs(stub.stub_host_sample(_p(f[1]), N, 997) - (r1[:N:997, 1, 0, 0].sum() + r1[:N:997, 2, 1, 1].sum())) < 1e-9
    assert f[1][N // 2 + 3, 2, 1, 0] == pytest.approx(r1[N // 2 + 3, 2, 1, 0], rel=1e-13)
    # the host changes the INPUT in place (a store to a write-protected page): the next call must see it
    stub.stub_host_scale(_p(f[0]), 5, 3.0)
    f[0][N - 1] *= -2.0                                      # ... and numpy stores, in the last (shared) page of the field
    k[5] *= 3.0; k[N - 1] *= -2.0
    assert np.array_equal(f[0], k[:N])
    d.Hopping_Matrix(0, _p(f[1]), _p(f[0]))
    orc.Hopping_Matrix(0, r1, k)
    assert rel_err(f[1], r1[:N]) < TOL
    # the host overwrites a stale OUTPUT array completely, then uses it as an input
    d.Hopping_Matrix(1, _p(f[2]), _p(f[1]))                  # f2 stale on the host
    f[2][:] = src[::-1]
    d.Hopping_Matrix(0, _p(f[3]), _p(f[2]))
    k2 = orc.new_field(); k2[:N] = src[::-1]
    r3 = orc.new_field(); orc.Hopping_Matrix(0, r3, k2)
    assert rel_err(f[3], r3[:N]) < TOL
    # neighbours share pages: f3's first page holds the end of f2, f3's last the start of f4 -- none of them was disturbed
    assert np.array_equal(f[2], src[::-1]) and not f[4].any()
    # in-place operator and a reduction on a stale array
    d.Qtm_pm_psi(_p(f[4]), _p(f[3]))
    q = orc.new_field(); orc.op("Qtm_pm_psi", q, r3)
    assert abs(d.square_norm(_p(f[4]), N, 1) - orc.square_norm(q, N)) < 1e-12 * orc.square_norm(q, N)
    assert rel_err(f[4], q[:N]) < TOL


def test_host_threads_read_a_stale_field_at_the_same_time(prog):
    stub, d, orc, f, (T, L, V, N), _ = prog
    stub.stub_host_sum_threads.restype = C.c_double; stub.stub_host_sum_threads.argtypes = [VP, C.c_int, C.c_int]
    d.tmlqcd_hip_set_residency(LAZY)
    src = random_spinor(74, N)
    f[0][:] = src
    k = orc.new_field(); k[:N] = src
    r1 = orc.new_field()
    for rep in range(3):
        d.Hopping_Matrix(rep & 1, _p(f[1]), _p(f[0]))
        orc.Hopping_Matrix(rep & 1, r1, k)
        want = float((r1[:N, :, :, 0] - r1[:N, :, :, 1]).sum())
        got = stub.stub_host_sum_threads(_p(f[1]), N, 8)      # eight threads fault into eight different places of the array
        assert abs(got - want) < 1e-9 * max(1.0, abs(want))


def test_full_lattice_fields_and_site_prefixes(prog):
    """Mirrors of the other two shapes: a lexicographic full-lattice field (D_psi: fetched whole on the first fault) and the
    first-N-sites prefix of an array (linalg with N < VOLUME/2); the same host bytes are never mirrored twice."""
    stub, d, orc, f, (T, L, V, N), block = prog
    d.D_psi.argtypes = [VP, VP]
    d.assign_add_mul_r.argtypes = [VP, VP, C.c_double, C.c_int]
    d.tmlqcd_hip_set_residency(LAZY)
    # two full fields = f[0..1] and f[2..3] taken together (fields lie back to back in the block)
    Pq = np.frombuffer(block, dtype=np.float64, count=V * 24, offset=f[0].ctypes.data - block.ctypes.data).reshape(V, 4, 3, 2)
    Pp = np.frombuffer(block, dtype=np.float64, count=V * 24, offset=f[2].ctypes.data - block.ctypes.data).reshape(V, 4, 3, 2)
    src = random_spinor(75, V)
    Pq[:] = src
    d.D_psi(_p(Pp), _p(Pq))
    q = orc.new_field(orc.VPR)
    q[:V] = src
    want = np.zeros_like(q)
    orc.D_psi(want, q)
    assert rel_err(Pp, want[:V]) < TOL                       # numpy loads fault into the stale full field
    # now address the first half of the output as a one-parity array: the full mirror gives way, the bytes stay right
    a = f[2].copy()
    d.Hopping_Matrix(0, _p(f[4]), _p(f[2]))
    k = orc.new_field(); k[:N] = a
    r = orc.new_field(); orc.Hopping_Matrix(0, r, k)
    assert rel_err(f[4], r[:N]) < TOL and np.array_equal(f[2], a) and np.array_equal(Pp[:N], a)
    # prefix of 1000 sites: P += c Q on the device, the rest of the array untouched and still readable
    y0 = f[4].copy()
    d.assign_add_mul_r(_p(f[4]), _p(f[2]), 0.5, 1000)
    assert rel_err(f[4][:1000], y0[:1000] + 0.5 * a[:1000]) < TOL and np.array_equal(f[4][1000:], y0[1000:])


def test_solver_and_mode_switch(prog):
    stub, d, orc, f, (T, L, V, N), _ = prog
    d.tmlqcd_hip_set_residency(LAZY)
    b = random_spinor(73, N)
    f[0][:] = b; f[1][:] = 0
    its = d.cg_her(_p(f[1]), _p(f[0]), 1000, 1e-20, 1, N, C.cast(d.Qtm_pm_psi, VP))
    assert its > 5
    d.tmlqcd_hip_set_residency(COHERENT)                     # leaving lazy mode: every host array current and writable again
    x = orc.new_field(); x[:N] = f[1]
    ax = orc.new_field(); orc.op("Qtm_pm_psi", ax, x)
    assert np.linalg.norm((ax[:N] - b).ravel()) < 1e-9 * np.linalg.norm(b.ravel())
    f[1][:] = 1.0                                            # plain stores, no fault handling involved any more
    assert f[1].sum() == N * 24
