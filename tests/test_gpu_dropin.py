"""GPU: the drop-in layer (reference symbol names, host AoS pointers, tmLQCD globals read at call time)."""
import ctypes as C

import numpy as np
import pytest

from tests.util import TOL, random_gauge, random_spinor, rel_err

pytestmark = pytest.mark.gpu
VP = C.c_void_p


def _p(a):
    return a.ctypes.data_as(VP)


@pytest.fixture(scope="module")
def host(host_stub):
    from oracle.oraclebind import Oracle
    stub, d = host_stub
    T, L = 8, 8
    kappa, mu, theta = 0.129, 0.013, (1.0, 0.0, 0.0, 0.0)
    V = T * L ** 3
    gptr = stub.stub_init(T, L, L, L)
    g = random_gauge(51, V)
    C.memmove(gptr, _p(g), g.nbytes)
    stub.stub_boundary(kappa, *theta)
    stub.stub_set_mu(mu)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=4)
    orc.set_gauge(g)
    d.Hopping_Matrix.argtypes = [C.c_int, VP, VP]
    d.tm_times_Hopping_Matrix.argtypes = [C.c_int, VP, VP, C.c_double, C.c_double]   # complex by value = (re, im) in SSE regs
    d.tm_sub_Hopping_Matrix.argtypes = [C.c_int, VP, VP, VP, C.c_double, C.c_double]
    for n in ("Qtm_pm_psi", "Qtm_minus_psi", "D_psi", "Q_pm_psi"):
        getattr(d, n).argtypes = [VP, VP]
    d.square_norm.restype = C.c_double; d.square_norm.argtypes = [VP, C.c_int, C.c_int]
    d.scalar_prod_r.restype = C.c_double; d.scalar_prod_r.argtypes = [VP, VP, C.c_int, C.c_int]
    d.assign_add_mul_r.argtypes = [VP, VP, C.c_double, C.c_int]
    d.add.argtypes = [VP, VP, VP, C.c_int]
    d.mul_r.argtypes = [VP, C.c_double, VP, C.c_int]
    d.cg_her.restype = C.c_int; d.cg_her.argtypes = [VP, VP, C.c_int, C.c_double, C.c_int, C.c_int, VP]
    d.tmlqcd_hip_set_residency.argtypes = [C.c_int]
    d.tmlqcd_hip_sync_to_host.argtypes = [VP]
    d.tmlqcd_hip_host_modified.argtypes = [VP]
    d.tmlqcd_hip_benchmark_loop.restype = C.c_double
    d.tmlqcd_hip_benchmark_loop.argtypes = [VP, VP, VP, C.c_int]
    yield stub, d, orc, g, (T, L, V)
    d.tmlqcd_hip_finalize()


def test_coherent_mode_is_a_plain_drop_in(host):
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    k = random_spinor(1, N); l = np.zeros_like(k); ref = orc.new_field()
    d.Hopping_Matrix(0, _p(l), _p(k)); orc.Hopping_Matrix(0, ref, k)
    assert rel_err(l, ref[:N]) < TOL
    assert stub.stub_gauge_flag() == 0                     # consumed like update_backward_gauge.c:240
    c = 0.6 + 0.2j
    p = random_spinor(2, N)
    d.tm_times_Hopping_Matrix(1, _p(l), _p(k), c.real, c.imag); orc.tm_times_Hopping_Matrix(1, ref, k, c)
    assert rel_err(l, ref[:N]) < TOL
    d.tm_sub_Hopping_Matrix(1, _p(l), _p(p), _p(k), c.real, c.imag); orc.tm_sub_Hopping_Matrix(1, ref, p, k, c)
    assert rel_err(l, ref[:N]) < TOL
    d.Qtm_pm_psi(_p(l), _p(k)); orc.op("Qtm_pm_psi", ref, k.copy())
    assert rel_err(l, ref[:N]) < TOL
    assert abs(d.square_norm(_p(k), N, 0) - orc.square_norm(k, N)) <= TOL * orc.square_norm(k, N)
    a = k.copy(); d.assign_add_mul_r(_p(a), _p(p), 0.5, N)
    assert rel_err(a, k + 0.5 * p) < TOL
    s = np.zeros_like(k); d.add(_p(s), _p(k), _p(p), N)          # linalg/add.c, linalg/mul_r.c (mixed_cg_her.c, operator.c callers)
    assert np.array_equal(s, k + p)
    d.mul_r(_p(s), 0.25, _p(k), N)
    assert np.array_equal(s, 0.25 * k)
    full = random_spinor(3, V); sf = np.zeros_like(full); d.mul_r(_p(sf), -2.0, _p(full), V)   # N = VOLUME: both parities
    assert np.array_equal(sf, -2.0 * full)
    kk = k.copy(); ref2 = orc.new_field(); ref2[:N] = k
    d.Qtm_minus_psi(_p(kk), _p(kk)); orc.op("Qtm_minus_psi", ref2, ref2)   # in place (invert_eo.c:270)
    assert rel_err(kk, ref2[:N]) < TOL


def test_fp32_site_diagonal_twists_on_blocks_of_any_length(host):
    """mul_one_pm_imu_inv_32 / assign_mul_one_pm_imu_inv_32 / mul_one_pm_imu_sub_mul_32 (the fp32 instances tm_operators.c
    generates from mul_one_pm_imu_inv_body.c / mul_one_pm_imu_sub_mul_body.c): solver/Msap.c calls them on domain blocks, so N is
    arbitrary.  Checked against the formula of the body files in float arithmetic."""
    stub, d, orc, g, (T, L, V) = host
    mu = np.float32(stub.stub_get_mu())
    d.mul_one_pm_imu_inv_32.argtypes = [VP, C.c_double, C.c_int]
    d.assign_mul_one_pm_imu_inv_32.argtypes = [VP, VP, C.c_double, C.c_int]
    d.mul_one_pm_imu_sub_mul_32.argtypes = [VP, VP, VP, C.c_double, C.c_int]

    def twist(k, z):                                            # spin 0,1: z, spin 2,3: conj(z)   ([N][4][3] complex64)
        out = k.copy()
        out[:, :2] = (z * k[:, :2]).astype(np.complex64); out[:, 2:] = (np.conj(z) * k[:, 2:]).astype(np.complex64)
        return out

    rng = np.random.default_rng(77)
    for N in (1, 100, 257, V // 2):
        k = (rng.standard_normal((N, 4, 3)) + 1j * rng.standard_normal((N, 4, 3))).astype(np.complex64)
        j = (rng.standard_normal((N, 4, 3)) + 1j * rng.standard_normal((N, 4, 3))).astype(np.complex64)
        for sign in (+1.0, -1.0):
            nrm = np.float32(1.0 / (1.0 + float(mu) ** 2))
            zinv = np.complex64(nrm + 1j * (1.0 if sign < 0 else -1.0) * nrm * mu)
            l = np.zeros_like(k)
            d.assign_mul_one_pm_imu_inv_32(_p(l), _p(k), sign, N)
            assert np.allclose(l, twist(k, zinv), rtol=2e-6, atol=1e-6)
            m = k.copy(); d.mul_one_pm_imu_inv_32(_p(m), sign, N)                       # in place
            assert np.allclose(m, twist(k, zinv), rtol=2e-6, atol=1e-6)
            z = np.complex64(1.0 + 1j * (1.0 if sign >= 0 else -1.0) * mu)
            d.mul_one_pm_imu_sub_mul_32(_p(l), _p(k), _p(j), sign, N)
            assert np.allclose(l, twist(k, z) - j, rtol=2e-6, atol=1e-6)
    d.mul_one_pm_imu_inv_32(_p(k), 1.0, 0)                      # N = 0: nothing happens


def test_linalg_on_any_site_count_and_the_reference_known_answers(host):
    """The reference's linalg loops over ANY N (its own unit test, tests/test_linalg_spinor.c, uses N = 2 and N = 1000 on arrays
    that are no lattice fields at all): the drop-in symbols take 0 <= N <= VOLUME, and reproduce the literal known answers."""
    import json, os
    stub, d, orc, g, (T, L, V) = host
    ka = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "linalg_known_answers.json")))
    d.diff.argtypes = [VP, VP, VP, C.c_int]
    d.assign.argtypes = [VP, VP, C.c_int]
    d.assign_mul_add_r.argtypes = [VP, C.c_double, VP, C.c_int]
    d.assign_mul_add_r_and_square.restype = C.c_double; d.assign_mul_add_r_and_square.argtypes = [VP, C.c_double, VP, C.c_int, C.c_int]
    d.gamma5.argtypes = [VP, VP, C.c_int]
    EPS = 1e-12

    def sp(rows):
        return np.ascontiguousarray(np.array(rows, dtype=np.float64).reshape(len(rows), 4, 3, 2))
    R, S = sp(ka["R"]), sp(ka["S"])
    assert abs(d.scalar_prod_r(_p(R), _p(S), 2, 0) - ka["scalar_prod_r_R_S"]) < EPS            # test_linalg_spinor.c:75-77
    assert abs(d.square_norm(_p(R), 2, 0) - ka["square_norm_R"]) < EPS                         # :157-160
    Q = np.zeros_like(R); d.diff(_p(Q), _p(R), _p(S), 2)                                       # :241-247
    assert abs(d.square_norm(_p(Q), 2, 0) - ka["diff_R_minus_S_norm"]) < EPS
    assert abs(Q[0, 0, 0, 0] - ka["diff_Q0_s0c0_re"]) < EPS and abs(Q[1, 2, 1, 1] - ka["diff_Q1_s2c1_im"]) < EPS
    A = R.copy(); d.assign_add_mul_r(_p(A), _p(S), ka["c"], 2)                                 # :328-334
    assert abs(d.square_norm(_p(A), 2, 0) - ka["assign_add_mul_r_norm"]) < EPS
    assert abs(A[0, 0, 0, 0] - ka["assign_add_mul_r_R0_s0c0_re"]) < EPS and abs(A[1, 2, 1, 1] - ka["assign_add_mul_r_R1_s2c1_im"]) < EPS
    B = R.copy(); d.assign_mul_add_r(_p(B), ka["c"], _p(S), 2)                                 # :414-420
    assert abs(d.square_norm(_p(B), 2, 0) - ka["assign_mul_add_r_norm"]) < EPS
    assert abs(B[0, 0, 0, 0] - ka["assign_mul_add_r_R0_s0c0_re"]) < EPS and abs(B[1, 2, 1, 1] - ka["assign_mul_add_r_R1_s2c1_im"]) < EPS
    # the timing half of the reference's test: N = 1000 random spinors; and a prefix longer than VOLUME/2; the SAME host buffer
    # with a different N right after (a mirror of another length must not be reused)
    rng = np.random.default_rng(5)
    for N in (1000, V // 2 + 100, 1, V - 1):
        a, b = rng.random((N, 4, 3, 2)), rng.random((N, 4, 3, 2))
        assert abs(d.scalar_prod_r(_p(a), _p(b), N, 0) - (a * b).sum()) <= 1e-13 * (a * b).sum()
        assert abs(d.square_norm(_p(a), N, 0) - (a * a).sum()) <= 1e-13 * (a * a).sum()
        assert abs(d.square_norm(_p(a), N // 2 + 1, 0) - (a[:N // 2 + 1] ** 2).sum()) <= 1e-13 * (a * a).sum()
        c = a.copy(); d.assign_add_mul_r(_p(c), _p(b), -0.3, N)
        assert rel_err(c, a - 0.3 * b) < TOL
        q = np.zeros((N + 1, 4, 3, 2)); q[N] = 7.0
        d.diff(_p(q), _p(a), _p(b), N)
        assert np.array_equal(q[:N], a - b) and np.all(q[N] == 7.0)                           # nothing written past N
        e = a.copy(); n2 = d.assign_mul_add_r_and_square(_p(e), 0.5, _p(b), N, 0)
        assert rel_err(e, 0.5 * a + b) < TOL and abs(n2 - (e * e).sum()) <= 1e-13 * n2
        g5 = np.zeros_like(a); d.gamma5(_p(g5), _p(a), N)
        assert np.array_equal(g5[:, :2], a[:, :2]) and np.array_equal(g5[:, 2:], -a[:, 2:])
    z = rng.random((4, 4, 3, 2)); z0 = z.copy()
    assert d.square_norm(_p(z), 0, 0) == 0.0 and d.scalar_prod_r(_p(z), _p(z), 0, 0) == 0.0    # N = 0: empty loops
    d.assign_add_mul_r(_p(z), _p(z), 2.0, 0); d.diff(_p(z), _p(z), _p(z), 0)
    assert np.array_equal(z, z0)


def test_mirror_registry_is_bounded(host):
    """Ever new host addresses (work fields allocated per solve, solver/solver_field.c): the registry drops the least recently
    used mirrors whose host copy is current instead of growing; results are unaffected."""
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    d.tmlqcd_hip_set_max_mirrors.argtypes = [C.c_int]
    d.tmlqcd_hip_set_max_mirrors(8)
    ref = orc.new_field()
    keep = []
    for i in range(30):
        k = random_spinor(100 + i, N); l = np.zeros_like(k)
        keep.append((k, l))                                   # distinct live addresses
        d.Hopping_Matrix(i & 1, _p(l), _p(k)); orc.Hopping_Matrix(i & 1, ref, k)
        assert rel_err(l, ref[:N]) < TOL, i
    k0, l0 = keep[0]                                          # its mirror is long gone: transparently re-created
    d.Hopping_Matrix(0, _p(l0), _p(k0)); orc.Hopping_Matrix(0, ref, k0)
    assert rel_err(l0, ref[:N]) < TOL
    # a host buffer first seen as a full-lattice field, then re-used (same address) as a one-parity input while the registry is
    # crowded: the re-created mirror must not become the eviction victim of the output mirror created right after it
    buf = random_spinor(200, V); tmp = np.zeros_like(buf)
    d.mul_r(_p(tmp), 2.0, _p(buf), V)
    lnew = np.zeros((N, 4, 3, 2))
    d.Hopping_Matrix(1, _p(lnew), _p(buf)); orc.Hopping_Matrix(1, ref, np.ascontiguousarray(buf[:N]))
    assert rel_err(lnew, ref[:N]) < TOL
    d.tmlqcd_hip_set_max_mirrors(64)


def test_globals_are_reread_at_call_time(host):
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    k = random_spinor(3, N); l = np.zeros_like(k); ref = orc.new_field()
    stub.stub_set_mu(-0.05); orc.set_mu(-0.05)             # callers flip g_mu around calls (tm_operators.c:382-385)
    stub.stub_boundary(0.11, 0.5, 0.0, 0.0, 0.25); orc.set_kappa_theta(0.11, (0.5, 0.0, 0.0, 0.25))
    d.Qtm_pm_psi(_p(l), _p(k)); orc.op("Qtm_pm_psi", ref, k.copy())
    assert rel_err(l, ref[:N]) < TOL
    g2 = random_gauge(52, V)
    C.memmove(stub.stub_init(T, L, L, L), _p(g2), g2.nbytes)   # new configuration + g_update_gauge_copy = 1
    orc.set_gauge(g2)
    d.Hopping_Matrix(1, _p(l), _p(k)); orc.Hopping_Matrix(1, ref, k)
    assert rel_err(l, ref[:N]) < TOL
    stub.stub_set_mu(0.013); orc.set_mu(0.013)
    stub.stub_boundary(0.129, 1.0, 0.0, 0.0, 0.0); orc.set_kappa_theta(0.129, (1.0, 0.0, 0.0, 0.0))


def test_full_lattice_operators(host):
    stub, d, orc, g, (T, L, V) = host
    lex = random_spinor(4, V); out = np.zeros_like(lex); ref = np.zeros_like(lex)
    d.D_psi(_p(out), _p(lex)); orc.D_psi(ref, lex)
    assert rel_err(out, ref) < TOL
    # Q_pm_psi = g5 D(+mu) g5 D(-mu)   (tm_operators.c:380-388)
    d.Q_pm_psi(_p(out), _p(lex))
    mu = orc.mu
    t1 = np.zeros_like(lex); t2 = np.zeros_like(lex)
    orc.set_mu(-mu); orc.D_psi(t1, lex); orc.gamma5(t2, t1, V); orc.set_mu(mu); orc.D_psi(t1, t2); orc.gamma5(ref, t1, V)
    assert rel_err(out, ref) < TOL
    # Q_pm_psi_prec (tm_operators.c:402) without the FFT preconditioner linked (weak references unresolved) is Q_pm_psi on a copy of k
    d.Q_pm_psi_prec.argtypes = [VP, VP]
    out2 = np.zeros_like(lex); keep = lex.copy()
    d.Q_pm_psi_prec(_p(out2), _p(lex))
    assert np.array_equal(lex, keep) and rel_err(out2, ref) < TOL and stub.stub_get_mu() == mu
    assert abs(d.square_norm(_p(lex), V, 0) - (lex ** 2).sum()) <= 1e-12 * (lex ** 2).sum()


def test_cg_her_drop_in_and_resident_benchmark(host):
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    q = random_spinor(5, N); P = np.zeros_like(q)
    f = C.cast(d.Qtm_pm_psi, VP)
    it = d.cg_her(_p(P), _p(q), 500, 1e-18, 1, N, f)
    Pref = orc.new_field(); it_ref, _ = orc.cg_her(Pref, q.copy(), 500, 1e-18, 1, N)
    assert abs(it - it_ref) <= 1 and rel_err(P, Pref[:N]) < 1e-8
    # resident mode: outputs stay in HBM until asked for
    f0 = random_spinor(6, N); f1 = np.zeros_like(f0); f2 = np.zeros_like(f0)
    d.tmlqcd_hip_set_residency(1)
    for a in (f0, f1, f2):                     # resident-mode contract: tell the library about host-side writes
        d.tmlqcd_hip_host_modified(_p(a))
    secs = d.tmlqcd_hip_benchmark_loop(_p(f0), _p(f1), _p(f2), 5)
    assert secs > 0 and not f2.any()                      # host copy untouched so far
    d.tmlqcd_hip_sync_to_host(_p(f2))
    r1, r2 = orc.new_field(), orc.new_field()
    orc.Hopping_Matrix(0, r1, f0); orc.Hopping_Matrix(1, r2, r1)
    assert rel_err(f2, r2[:N]) < TOL
    d.tmlqcd_hip_set_residency(0)


class SolverParams(C.Structure):
    """solver_params_t (solver/solver_params.h:46-109): 144 bytes, mcg_delta at offset 52."""
    _fields_ = [("eigcg_i", C.c_int * 5), ("eigcg_d", C.c_double * 3), ("eigcg_rand_guess_opt", C.c_int), ("mcg_delta", C.c_float),
                ("type", C.c_int), ("max_iter", C.c_int), ("rel_prec", C.c_int), ("no_shifts", C.c_int), ("sdim", C.c_int),
                ("squared_solver_prec", C.c_double), ("M_psi", VP), ("M_psi32", VP), ("M_ndpsi", VP), ("M_ndpsi32", VP),
                ("shifts", VP), ("solution_type", C.c_int), ("compression_type", C.c_int), ("sloppy_precision", C.c_int),
                ("external_inverter", C.c_int)]


def test_mixed_cg_her_and_rg_mixed_cg_her_drop_in(host):
    """solver/mixed_cg_her.h, solver/rg_mixed_cg_her.h signatures: solver_params_t BY VALUE (>16 B => passed in memory),
    f32 on the stack behind it."""
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    assert C.sizeof(SolverParams) == 144 and SolverParams.mcg_delta.offset == 52
    q = random_spinor(8, N)
    for name, delta in (("mixed_cg_her", 0.0), ("rg_mixed_cg_her", 0.1)):
        fn = getattr(d, name)
        fn.restype = C.c_int
        fn.argtypes = [VP, VP, SolverParams, C.c_int, C.c_double, C.c_int, C.c_int, VP, VP]
        sp = SolverParams(); sp.mcg_delta = delta
        P = np.full_like(q, 2.0)                                     # both solvers start from zero themselves
        it = fn(_p(P), _p(q), sp, 5000, 1e-20, 1, N, C.cast(d.Qtm_pm_psi, VP), VP(0xdead0))
        assert it > 0, name
        full = orc.new_field(); full[:N] = P
        chk = orc.new_field(); orc.op("Qtm_pm_psi", chk, full)
        assert ((chk[:N] - q) ** 2).sum() / (q ** 2).sum() <= 1e-20, name


def test_invert_eo_call_sequence_through_the_drop_in(host):
    """The even/odd inversion of invert_eo.c:143-148,296-310 statement by statement through the reference-named
    symbols on host arrays: coherent mode, then resident mode with one download at the end.  The result solves
    M_full (Even_new, Odd_new) = (Even, Odd) and equals the oracle running the same statements."""
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    d.assign_mul_one_pm_imu_inv.argtypes = [VP, VP, C.c_double, C.c_int]
    d.mul_one_pm_imu_inv.argtypes = [VP, C.c_double, C.c_int]
    d.assign_mul_add_r.argtypes = [VP, C.c_double, VP, C.c_int]
    d.gamma5.argtypes = [VP, VP, C.c_int]
    Even, Odd = random_spinor(61, N), random_spinor(62, N)
    prec, max_iter = 1e-22, 1000

    # the oracle, same statements
    o_en, o_on, o_d = orc.new_field(), orc.new_field(), orc.new_field()
    orc.assign_mul_one_pm_imu_inv(o_en, Even, +1., N)
    orc.Hopping_Matrix(1, o_d, o_en)
    orc.assign_mul_add_r(o_d, +1., Odd, N)
    orc.gamma5(o_d, o_d, N)
    it_ref, _ = orc.cg_her(o_on, o_d.copy(), max_iter, prec, 1, N)
    orc.op("Qtm_minus_psi", o_on, o_on)
    orc.Hopping_Matrix(0, o_d, o_on)
    orc.mul_one_pm_imu_inv(o_d, +1., N)
    orc.assign_add_mul_r(o_en, o_d, +1., N)

    for resident in (0, 1):
        en, on, dum = np.zeros_like(Even), np.zeros_like(Even), np.zeros_like(Even)
        d.tmlqcd_hip_set_residency(resident)
        if resident:
            for a in (Even, Odd, en, on, dum):
                d.tmlqcd_hip_host_modified(_p(a))
        d.assign_mul_one_pm_imu_inv(_p(en), _p(Even), +1., N)                    # invert_eo.c:143
        d.Hopping_Matrix(1, _p(dum), _p(en))                                     # :145  (OE)
        d.assign_mul_add_r(_p(dum), +1., _p(Odd), N)                             # :148
        d.gamma5(_p(dum), _p(dum), N)                                            # :296
        it = d.cg_her(_p(on), _p(dum), max_iter, prec, 1, N, C.cast(d.Qtm_pm_psi, VP))   # :302
        d.Qtm_minus_psi(_p(on), _p(on))                                          # :303
        d.Hopping_Matrix(0, _p(dum), _p(on))                                     # :306  (EO)
        d.mul_one_pm_imu_inv(_p(dum), +1., N)                                    # :307
        d.assign_add_mul_r(_p(en), _p(dum), +1., N)                              # :310
        if resident:
            assert not en.any() and not on.any()                                 # nothing came back yet
            d.tmlqcd_hip_sync_to_host(_p(en)); d.tmlqcd_hip_sync_to_host(_p(on))
        d.tmlqcd_hip_set_residency(0)
        assert abs(it - it_ref) <= 1
        assert rel_err(en, o_en[:N]) < 1e-8 and rel_err(on, o_on[:N]) < 1e-8
        # and it really is the solution of the full (unpreconditioned) system
        re, ro = orc.new_field(), orc.new_field()
        fe, fo = orc.new_field(), orc.new_field(); fe[:N] = en; fo[:N] = on
        orc.M_full(re, ro, fe, fo)
        res = ((re[:N] - Even) ** 2).sum() + ((ro[:N] - Odd) ** 2).sum()
        assert res / ((Even ** 2).sum() + (Odd ** 2).sum()) < 1e-18


def test_clover_drop_in(host):
    """operator/clovertm_operators.h symbols: sw / sw_inv are read from the host program's globals."""
    from tests.util import random_clover
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    mu = orc.mu
    sw, swi = random_clover(9, orc, mu, scale=0.05)
    stub.stub_init_clover.restype = C.c_void_p
    stub.stub_init_clover.argtypes = [C.c_int]
    C.memmove(stub.stub_init_clover(0), _p(sw), sw.nbytes)
    C.memmove(stub.stub_init_clover(1), _p(swi), swi.nbytes)
    d.tmlqcd_hip_update_clover()
    orc.set_clover(sw, swi)
    d.Qsw_pm_psi.argtypes = [VP, VP]
    d.clover_inv.argtypes = [VP, C.c_int, C.c_double]
    d.clover_gamma5.argtypes = [C.c_int, VP, VP, VP, C.c_double]
    k, j = random_spinor(11, N), random_spinor(12, N)
    l = np.zeros_like(k); ref = orc.new_field()
    d.Qsw_pm_psi(_p(l), _p(k)); orc.op("Qsw_pm_psi", ref, k.copy())
    assert rel_err(l, ref[:N]) < TOL
    stub.stub_set_mu3(0.05); orc.set_mu3(0.05)                        # g_mu3 is read at call time like g_mu
    try:
        d.Qsw_pm_psi(_p(l), _p(k)); orc.op("Qsw_pm_psi", ref, k.copy())
    finally:
        stub.stub_set_mu3(0.0); orc.set_mu3(0.0)
    assert rel_err(l, ref[:N]) < TOL
    kk = k.copy(); ref[:N] = k
    d.clover_inv(_p(kk), -1, mu); orc.clover_inv(ref, -1, mu)
    assert rel_err(kk, ref[:N]) < TOL
    d.clover_gamma5(1, _p(l), _p(k), _p(j), mu); orc.clover_gamma5(1, ref, k, j, mu)
    assert rel_err(l, ref[:N]) < TOL
    # D_psi with g_c_sw > 0 (D_psi_body.c:314-316): the site term is (1 + T + i mu g5) from the host's sw array, on both parities
    lex = random_spinor(14, V); out = np.zeros_like(lex)
    e2l = orc.eo2lexic()
    fe, fo, en, on = orc.new_field(), orc.new_field(), orc.new_field(), orc.new_field()
    fe[:N] = lex[e2l[:N]]; fo[:N] = lex[e2l[N:V]]
    orc.Msw_full(en, on, fe, fo)
    expect = np.zeros_like(lex); expect[e2l[:N]] = en[:N]; expect[e2l[N:V]] = on[:N]
    stub.stub_set_csw(1.0)
    try:
        d.D_psi(_p(out), _p(lex))
    finally:
        stub.stub_set_csw(0.0)
    assert rel_err(out, expect) < TOL
    d.D_psi(_p(out), _p(lex)); plain = np.zeros_like(lex); orc.D_psi(plain, lex)        # and back on the plain branch
    assert rel_err(out, plain) < TOL and rel_err(expect, plain) > 1e-3
    q = random_spinor(13, N); P = np.zeros_like(q)
    it = d.cg_her(_p(P), _p(q), 2000, 1e-18, 1, N, C.cast(d.Qsw_pm_psi, VP))
    Pref = orc.new_field(); it_ref, _ = orc.cg_her(Pref, q.copy(), 2000, 1e-18, 1, N, "Qsw_pm_psi")
    assert abs(it - it_ref) <= max(1, it_ref // 100) and rel_err(P, Pref[:N]) < 1e-7


def test_invert_clover_eo_call_sequence_through_the_drop_in(host):
    """invert_clover_eo.c:101-160 (BASELINE configs[4] call stack) statement by statement through the drop-in:
    sw_term / sw_invert computed on the device into the host program's sw / sw_inv arrays, then the even/odd inversion with
    cg_her, mixed_cg_her and rg_mixed_cg_her on Qsw_pm_psi and Qm = Qsw_minus_psi in place.  The result solves
    Msw_full (Even_new, Odd_new) = (Even, Odd)."""
    stub, d, orc, g, (T, L, V) = host
    N = V // 2
    kappa, mu, c_sw = 0.129, 0.013, 1.4
    stub.stub_boundary(kappa, 1.0, 0.0, 0.0, 0.0); stub.stub_set_mu(mu)
    orc.set_kappa_theta(kappa, (1.0, 0.0, 0.0, 0.0)); orc.set_mu(mu)
    gptr = stub.stub_init(T, L, L, L)            # fresh links: also exercises the re-upload
    C.memmove(gptr, _p(g), g.nbytes)
    orc.set_gauge(g)
    stub.stub_init_clover.restype = C.c_void_p
    stub.stub_init_clover.argtypes = [C.c_int]
    sw_host = np.frombuffer((C.c_double * (V * 6 * 18)).from_address(stub.stub_init_clover(0)), dtype=np.float64).reshape(V, 3, 2, 3, 3, 2)
    swi_host = np.frombuffer((C.c_double * (V * 8 * 18)).from_address(stub.stub_init_clover(1)), dtype=np.float64).reshape(V, 4, 2, 3, 3, 2)
    d.tmlqcd_hip_sw_term.argtypes = [C.c_double, C.c_double]
    d.tmlqcd_hip_sw_invert.argtypes = [C.c_int, C.c_double]
    d.tmlqcd_hip_sw_term(kappa, c_sw)                                  # operator.c:329-330
    d.tmlqcd_hip_sw_invert(0, mu)                                      # operator.c:364  sw_invert(EE, mu)
    sw = orc.sw_term(kappa, c_sw); swi, _ = orc.sw_invert(sw, 0, mu)
    assert rel_err(sw_host, sw) < TOL and rel_err(swi_host, swi) < TOL  # the host program's arrays received the copies
    orc.set_clover(sw, swi)
    for n in ("assign_mul_one_sw_pm_imu_inv", "assign_mul_one_sw_pm_imu"):
        getattr(d, n).argtypes = [C.c_int, VP, VP, C.c_double]
    d.assign_mul_add_r.argtypes = [VP, C.c_double, VP, C.c_int]
    d.gamma5.argtypes = [VP, VP, C.c_int]
    d.clover_inv.argtypes = [VP, C.c_int, C.c_double]
    d.Qsw_minus_psi.argtypes = [VP, VP]
    for n in ("mixed_cg_her", "rg_mixed_cg_her"):
        getattr(d, n).restype = C.c_int
        getattr(d, n).argtypes = [VP, VP, SolverParams, C.c_int, C.c_double, C.c_int, C.c_int, VP, VP]
    Even, Odd = random_spinor(71, N), random_spinor(72, N)
    prec, max_iter = 1e-22, 2000
    Qsq = C.cast(d.Qsw_pm_psi, VP)
    for solver in ("CG", "MIXEDCG", "RGMIXEDCG"):
        en, on, dum = np.zeros_like(Even), np.zeros_like(Even), np.zeros_like(Even)
        d.assign_mul_one_sw_pm_imu_inv(0, _p(en), _p(Even), +mu)                 # invert_clover_eo.c:101
        d.Hopping_Matrix(1, _p(dum), _p(en))                                     # :103
        d.assign_mul_add_r(_p(dum), +1., _p(Odd), N)                             # :106
        d.gamma5(_p(dum), _p(dum), N)                                            # :125 / :141 / :148
        sp = SolverParams(); sp.mcg_delta = 0.1
        if solver == "CG":
            it = d.cg_her(_p(on), _p(dum), max_iter, prec, 1, N, Qsq)            # :126-128
        elif solver == "MIXEDCG":
            it = d.mixed_cg_her(_p(on), _p(dum), sp, max_iter, prec, 1, N, Qsq, None)        # :142-144
        else:
            it = d.rg_mixed_cg_her(_p(on), _p(dum), sp, max_iter, prec, 1, N, Qsq, None)     # :149-150
        assert it > 0, solver
        d.Qsw_minus_psi(_p(on), _p(on))                                          # Qm(Odd_new, Odd_new)
        d.Hopping_Matrix(0, _p(dum), _p(on))                                     # :159
        d.clover_inv(_p(dum), +1, mu)                                            # :160
        d.assign_add_mul_r(_p(en), _p(dum), +1., N)                              # :163
        fe, fo, re, ro = orc.new_field(), orc.new_field(), orc.new_field(), orc.new_field()
        fe[:N] = en; fo[:N] = on
        orc.Msw_full(re, ro, fe, fo)
        res = ((re[:N] - Even) ** 2).sum() + ((ro[:N] - Odd) ** 2).sum()
        assert res / ((Even ** 2).sum() + (Odd ** 2).sum()) < 1e-18, solver
    # Msw_full through the drop-in as well
    d.Msw_full.argtypes = [VP] * 4
    a, b = np.zeros_like(Even), np.zeros_like(Even)
    d.Msw_full(_p(a), _p(b), _p(Even), _p(Odd))
    ra, rb = orc.new_field(), orc.new_field()
    orc.Msw_full(ra, rb, Even, Odd)
    assert rel_err(a, ra[:N]) < TOL and rel_err(b, rb[:N]) < TOL
