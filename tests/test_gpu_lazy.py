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
_LAZY_BLOCKS = 0


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
    # The block lives in a mapping of its own at an address range nothing in this process has used before: a range that once held an
    # array the HIP runtime copied from / to directly (the big pageable numpy arrays of the full-size tests that run earlier in the
    # same process) stays registered with the driver after it has been freed, and every mprotect on a reused part of it then goes
    # through the driver's MMU notifier -- 28 ms per call instead of microseconds.  A tmLQCD process allocates its
    # spinor fields once and, in lazy mode, never lets the runtime touch them: that is the situation this fixture reproduces.
    import os
    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
    nbytes = (nf * N * 192 + 4096 + 64 + 4095) // 4096 * 4096
    global _LAZY_BLOCKS
    hint = 0x6a0000000000 + (os.getpid() % 4096) * (1 << 32) + _LAZY_BLOCKS * (1 << 30)
    _LAZY_BLOCKS += 1
    addr = libc.mmap(hint, nbytes, 3, 0x22, -1, 0)            # PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS (zero-filled)
    assert addr not in (None, C.c_void_p(-1).value), "mmap failed"
    block = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(addr))
    base = (block.ctypes.data + 31) // 32 * 32 + 32           # 32-byte aligned, NOT page aligned
    off = base - block.ctypes.data
    fields = [np.frombuffer(block, dtype=np.float64, count=N * 24, offset=off + i * N * 192).reshape(N, 4, 3, 2) for i in range(nf)]
    assert fields[0].ctypes.data % 4096 != 0
    yield stub, d, orc, fields, (T, L, V, N), block
    d.tmlqcd_hip_set_residency(COHERENT)
    d.tmlqcd_hip_finalize()
    del fields, block
    libc.munmap(addr, nbytes)


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
    # a prefix LONGER than VOLUME/2 (a block volume): two parts on the device, fetched whole into the bounce buffer and its second half
    M = N + 1000
    p0, q0 = Pp[:M].copy(), Pq[:M].copy()
    d.assign_add_mul_r(_p(Pp), _p(Pq), -0.25, M)
    assert rel_err(Pp[:M], p0 - 0.25 * q0) < TOL and np.array_equal(Pq[:M], q0)
    assert rel_err(Pp[M:N + 2000], want[M:N + 2000]) < TOL    # behind the prefix: D_psi's result, untouched


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


def test_halves_then_the_full_field_at_the_same_base(prog):
    """Hopping_Matrix into the two halves of a lexicographic field (two EO mirrors at X and X + V/2), then D_psi
    and square_norm(., VOLUME) on the pair (a FULL mirror / a prefix at X) -- the mirror of the odd half must give way, or the same
    host bytes have two independently valid device copies -- then an EO operation on the odd half again."""
    stub, d, orc, f, (T, L, V, N), block = prog
    d.D_psi.argtypes = [VP, VP]
    d.tmlqcd_hip_set_residency(LAZY)
    src = random_spinor(76, N)
    f[4][:] = src
    k = orc.new_field(); k[:N] = src
    r0, r1 = orc.new_field(), orc.new_field()
    d.Hopping_Matrix(1, _p(f[0]), _p(f[4])); orc.Hopping_Matrix(1, r0, k)        # even half of the pair (f0, f1) <- H_oe... written by the device
    d.Hopping_Matrix(0, _p(f[1]), _p(f[4])); orc.Hopping_Matrix(0, r1, k)        # odd half: a second EO mirror at X + V/2
    pair = np.frombuffer(block, dtype=np.float64, count=V * 24, offset=f[0].ctypes.data - block.ctypes.data).reshape(V, 4, 3, 2)
    outp = np.frombuffer(block, dtype=np.float64, count=V * 24, offset=f[2].ctypes.data - block.ctypes.data).reshape(V, 4, 3, 2)
    want_pair = np.concatenate([r0[:N], r1[:N]])
    nrm = d.square_norm(_p(f[0]), V, 1)                                           # a prefix of VOLUME sites at the base of the even half
    assert abs(nrm - (want_pair ** 2).sum()) <= 1e-12 * nrm
    d.D_psi(_p(outp), _p(pair))                                                   # FULL mirror at the same base
    q = orc.new_field(orc.VPR); q[:V] = want_pair
    want = np.zeros_like(q); orc.D_psi(want, q)
    assert rel_err(outp, want[:V]) < TOL
    assert rel_err(pair, want_pair) < TOL                                         # the host sees the halves the device wrote, not a stale page
    # the host changes the odd half, then an EO operation on it: must see the new data, not the old EO mirror's
    f[1][:] = 2.0 * r1[:N]
    d.Hopping_Matrix(1, _p(f[4]), _p(f[1]))
    k1 = orc.new_field(); k1[:N] = 2.0 * r1[:N]
    r4 = orc.new_field(); orc.Hopping_Matrix(1, r4, k1)
    assert rel_err(f[4], r4[:N]) < TOL


def test_upload_next_to_a_stale_full_field(prog):
    """the upload's host -> staging copy faults on the page it shares with a neighbouring FULL field whose host
    copy is stale; the handler fetches that whole field -- through a staging buffer of its own, not the one being filled."""
    stub, d, orc, f, (T, L, V, N), block = prog
    d.D_psi.argtypes = [VP, VP]
    d.tmlqcd_hip_set_residency(LAZY)
    pin = np.frombuffer(block, dtype=np.float64, count=V * 24, offset=f[0].ctypes.data - block.ctypes.data).reshape(V, 4, 3, 2)
    pout = np.frombuffer(block, dtype=np.float64, count=V * 24, offset=f[2].ctypes.data - block.ctypes.data).reshape(V, 4, 3, 2)
    src = random_spinor(77, V)
    pin[:] = src
    d.D_psi(_p(pout), _p(pin))                               # pout = (f2, f3): written by the device, stale on the host, its last page shared with f4
    q = orc.new_field(orc.VPR); q[:V] = src
    want = np.zeros_like(q); orc.D_psi(want, q)
    x = random_spinor(78, N)
    f[4][:] = x                                              # host-modified neighbour (its first page is pout's last: the store fetched pout? no -- only that page's owner decides)
    d.Hopping_Matrix(0, _p(f[0]), _p(f[4]))                  # upload of f4: memcpy reads the shared page
    k = orc.new_field(); k[:N] = x
    r = orc.new_field(); orc.Hopping_Matrix(0, r, k)
    assert rel_err(f[0], r[:N]) < TOL
    assert rel_err(pout, want[:V]) < TOL and np.array_equal(f[4], x)


def test_work_field_freed_and_allocated_again_between_calls(prog):
    """solver/solver_field.c allocates its work fields per solve -- a large calloc is unmapped by free and the
    next one gets the same address back, readable and writable, so no fault tells the library.  A watched mirror is therefore
    probed before it is trusted; the second round must compute from the NEW contents.  (The stub maps and unmaps the blocks itself,
    as glibc does for a block of this size that no free heap chunk can serve; a block recycled INSIDE the heap has to be announced
    with tmlqcd_hip_forget -- free() would write its bookkeeping into a protected page while holding the allocator's lock.)"""
    stub, d, orc, f, (T, L, V, N), _ = prog
    stub.stub_calloc.restype = VP; stub.stub_calloc.argtypes = [C.c_size_t]
    stub.stub_free.argtypes = [VP, C.c_size_t]
    d.tmlqcd_hip_set_residency(LAZY)
    nb = N * 192
    seen = set()
    same = 0
    for rnd in range(3):
        px, py = stub.stub_calloc(nb), stub.stub_calloc(nb)
        same += (px in seen) + (py in seen)
        seen.update((px, py))
        x = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_double)), shape=(N, 4, 3, 2))
        y = np.ctypeslib.as_array(C.cast(py, C.POINTER(C.c_double)), shape=(N, 4, 3, 2))
        src = random_spinor(80 + rnd, N)
        x[:] = src
        d.Hopping_Matrix(0, py, px)                          # x uploaded (write-protected), y stale on the host (inaccessible)
        k = orc.new_field(); k[:N] = src
        r = orc.new_field(); orc.Hopping_Matrix(0, r, k)
        if rnd == 1:
            assert rel_err(y, r[:N]) < TOL                   # (one round reads the result, the others free it unread)
        nrm = d.square_norm(py, N, 1)
        assert abs(nrm - orc.square_norm(r, N)) <= 1e-12 * nrm, rnd
        del x, y
        stub.stub_free(py, nb); stub.stub_free(px, nb)
    assert same >= 2, "the allocator did not hand the same addresses out again: the test did not exercise the probe"


def test_threads_fault_while_the_master_thread_changes_the_registry(prog):
    """host threads read a stale field WHILE the master thread issues drop-in calls on other fields (an insert
    and an erase of the registry per call): 100 rounds, no crash, every sum equal to the oracle's."""
    stub, d, orc, f, (T, L, V, N), _ = prog
    fn = stub.stub_threads_read_while_master_calls
    fn.restype = C.c_double
    fn.argtypes = [VP, VP, VP, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int]
    d.tmlqcd_hip_forget.argtypes = [VP]
    d.tmlqcd_hip_set_residency(LAZY)
    src = random_spinor(79, N)
    f[0][:] = src
    k = orc.new_field(); k[:N] = src
    outs = [np.zeros((N, 4, 3, 2)) for _ in range(6)]
    arr = (VP * 6)(*[o.ctypes.data for o in outs])
    r = [orc.new_field(), orc.new_field()]
    for ieo in (0, 1):
        orc.Hopping_Matrix(ieo, r[ieo], k)
    want = [float((r[i][:N, :, :, 0] - r[i][:N, :, :, 1]).sum()) for i in (0, 1)]
    for rnd in range(100):
        d.Hopping_Matrix(rnd & 1, _p(f[1]), _p(f[0]))        # f1 stale on the host
        got = fn(C.cast(d.Hopping_Matrix, VP), C.cast(d.tmlqcd_hip_forget, VP), _p(f[1]), N, 8, _p(f[0]), arr, 6, 7)
        assert abs(got - want[rnd & 1]) < 1e-9 * max(1.0, abs(want[rnd & 1])), rnd
    d.Hopping_Matrix(0, _p(outs[0]), _p(f[0]))
    assert rel_err(outs[0], r[0][:N]) < TOL


def test_faults_of_the_host_program_reach_its_own_handler():
    """VERDICT r2 item 7 (chain test): a host program with a SIGSEGV handler of its own, installed BEFORE the lazy mode's -- its faults
    (a guard page) must still reach it while fields are stale and watched, and the library's own faults must be served as before.
    A process of its own: in this one the library's handler is long installed (tests/lazy_chain_child.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "lazy_chain_child.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().startswith("OK"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("where,why", [("heap", "inside the malloc heap"), ("arena", "inside a thread's malloc arena"), ("shared", "a shared mapping")])
def test_arrays_that_cannot_be_watched(where, why):
    """the hang of round 3 was a fault taken INSIDE malloc, on a watched page of the malloc heap.  Arrays in a
    malloc arena (main or per-thread) or in a shared mapping are recognised from /proc/self/maps and never watched: copied per call,
    results the coherent mode's, no fault served.  And with the test hook that watches them anyway, the first fault ends the process
    with a message within seconds -- it fails loudly, it does not hang (tests/lazy_unwatchable_child.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = [sys.executable, os.path.join(root, "tests", "lazy_unwatchable_child.py"), where]
    env = dict(os.environ, TMLQCD_HIP_LAZY_DEBUG="1")
    env.pop("TMLQCD_HIP_LAZY_FORCE_WATCH", None)
    r = subprocess.run(child, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().startswith("OK") and r.stdout.strip().endswith("faults served 0"), (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "not watched, copied per call: " + why in r.stderr, r.stderr[-1500:]
    r = subprocess.run(child, env=dict(env, TMLQCD_HIP_LAZY_FORCE_WATCH="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 1 and "must not be watched" in r.stderr and "OK" not in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
