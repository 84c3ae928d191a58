"""GPU: parity cases VERDICT r1 listed as missing.

1. The reference's only held golden vectors for this path (tests/test_linalg_spinor.c:75-420, N = 2) through libtmlqcd_hip.so.
2. Hopping_Matrix_nocom (operator/Hopping_Matrix_nocom.c:48-56): unsplit == Hopping_Matrix; on a split lattice == interior +
   boundary kernels on whatever faces the last exchange left behind.
3. BASELINE configs[3]'s real shape: 32^3 x 64 over EIGHT slabs of T_local = 8, slab by slab, and the fused CG iteration on a
   T_local = 8 rank (self-exchange) against the unsplit solve.
4. mixed_cg_her's iteration counts and restart points against solver/mixed_cg_her.c:65-202 restated over the reference's own
   object code (oracle/mixed_cg_ref.py, fixture tests/golden/ref_mixed_*).
5. tmhip_multi_hopping_matrix back to back with one rank held back (write-after-read on the receive buffers).
"""
import json
import os

import numpy as np
import pytest

from tests.util import TOL, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _sp(rows):
    a = np.zeros((len(rows), 4, 3, 2))
    for i, r in enumerate(rows):
        a[i] = np.array(r, dtype=np.float64).reshape(4, 3, 2)
    return a


def test_reference_linalg_known_answers_on_the_gpu():
    """tests/test_linalg_spinor.c: literal inputs -> literal results, N = 2 (a ragged prefix of a 4^4 field)."""
    from tmlqcd_amd import Lattice
    ka = json.load(open(os.path.join(GOLD, "linalg_known_answers.json")))
    EPS = 1e-12          # the reference: 1e-15 on value/1000, one-sided; here two-sided on the value itself
    lat = Lattice(4, 4, 4, 4)
    R, S = _sp(ka["R"]), _sp(ka["S"])
    dR, dS, dQ = lat.field(R), lat.field(S), lat.field()
    assert abs(lat.scalar_prod_r(dR, dS, 2) - ka["scalar_prod_r_R_S"]) < EPS        # :75-77
    assert abs(lat.square_norm(dR, 2) - ka["square_norm_R"]) < EPS                  # :157-160
    lat.diff(dQ, dR, dS, 2)                                                         # :241-247
    Q = dQ.download(2)
    assert abs(lat.square_norm(dQ, 2) - ka["diff_R_minus_S_norm"]) < EPS
    assert abs(Q[0, 0, 0, 0] - ka["diff_Q0_s0c0_re"]) < EPS and abs(Q[1, 2, 1, 1] - ka["diff_Q1_s2c1_im"]) < EPS
    dA = lat.field(R)
    lat.assign_add_mul_r(dA, dS, ka["c"], 2)                                        # :328-334
    A = dA.download(2)
    assert abs(lat.square_norm(dA, 2) - ka["assign_add_mul_r_norm"]) < EPS
    assert abs(A[0, 0, 0, 0] - ka["assign_add_mul_r_R0_s0c0_re"]) < EPS and abs(A[1, 2, 1, 1] - ka["assign_add_mul_r_R1_s2c1_im"]) < EPS
    dB = lat.field(R)
    lat.assign_mul_add_r(dB, ka["c"], dS, 2)                                        # :414-420
    B = dB.download(2)
    assert abs(lat.square_norm(dB, 2) - ka["assign_mul_add_r_norm"]) < EPS
    assert abs(B[0, 0, 0, 0] - ka["assign_mul_add_r_R0_s0c0_re"]) < EPS and abs(B[1, 2, 1, 1] - ka["assign_mul_add_r_R1_s2c1_im"]) < EPS
    dC = lat.field(R)
    n = lat.assign_mul_add_r_and_square(dC, ka["c"], dS, 2)
    assert abs(n - ka["assign_mul_add_r_norm"]) < EPS and np.array_equal(dC.download(2), B)
    assert np.array_equal(dC.download(4)[2:], np.zeros((2, 4, 3, 2)))               # sites beyond N untouched
    lat.close()


@pytest.mark.parametrize("ieo", [0, 1])
def test_hopping_matrix_nocom(ieo):
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, L = 8, 8
    kappa, theta = 0.13, (1.0, 0.0, 0.0, 0.3)
    lat = Lattice(T, L, L, L, kappa=kappa, theta=theta)
    orc = Oracle(T, L, L, L, kappa=kappa, theta=theta, threads=4)
    g = syn.gauge_field(61, T, L, L, L)
    lat.set_gauge(g); orc.set_gauge(g)
    N, face = lat.Vh, L ** 3 // 2
    k1, k2 = syn.spinor_field_eo(62, 1 - ieo, T, L, L, L), syn.spinor_field_eo(63, 1 - ieo, T, L, L, L)
    d1, d2, dl = lat.field(k1), lat.field(k2), lat.field()

    def H(k):
        a, b = orc.new_field(), orc.new_field()
        a[:N] = k
        orc.Hopping_Matrix(ieo, b, a)
        return b[:N].copy()
    # unsplit lattice: no communication exists, nocom == Hopping_Matrix (Hopping_Matrix_nocom.c:48-56 is the same body)
    lat.Hopping_Matrix_nocom(ieo, dl, d2)
    assert rel_err(dl.download(), H(k2)) < TOL
    # split lattice (self-exchange): a communicating call leaves k1's faces in the receive buffers, the nocom call on k2 then
    # uses them for the two t-hops that cross the boundary -- everything else comes from k2.  By linearity the expected field is
    # H k2 + (crossing hops of k1 - k2): H of the difference restricted to slice T-1 seen from slice 0, and to slice 0 from T-1.
    for mode in (1, 2, 3):
        lat.set_loopback(mode)
        lat.Hopping_Matrix(ieo, dl, d1)
        assert rel_err(dl.download(), H(k1)) < TOL
        lat.Hopping_Matrix_nocom(ieo, dl, d2)
        exp = H(k2)
        dlast, dfirst = np.zeros_like(k1), np.zeros_like(k1)
        dlast[N - face:] = (k1 - k2)[N - face:]
        dfirst[:face] = (k1 - k2)[:face]
        exp[:face] += H(dlast)[:face]
        exp[N - face:] += H(dfirst)[N - face:]
        assert rel_err(dl.download(), exp) < TOL, mode
        lat.Hopping_Matrix(ieo, dl, d2)                  # and communication back on: the plain result
        assert rel_err(dl.download(), H(k2)) < TOL
    lat.close()


def _mem_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                return int(ln.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def test_t_split_32x64_over_eight_slabs_of_t8():
    """configs[3] as the 8-GPU node will run it: 32^3 x 64 cut into EIGHT slabs of T_local = 8 (eight contexts of this process,
    peer-copy ring; the same pack / interior / boundary kernels as the RCCL path) == the unsplit lattice on the oracle."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    from tmlqcd_amd.hip import multi_Hopping_Matrix
    if _mem_gb() < 24:
        pytest.skip("needs ~20 GB of host memory for the oracle's gauge copy")
    T, L, world = 8, 32, 8
    Tg = T * world
    kappa, theta = 0.125, (1.0, 0.0, 0.0, 0.0)
    g = Oracle(Tg, L, L, L, kappa=kappa, theta=theta, threads=16)
    g.set_gauge(syn.gauge_field(71, Tg, L, L, L))
    lats = [Lattice(T, L, L, L, kappa=kappa, theta=theta, nproc_t=world, proc_t=r) for r in range(world)]
    for r, lat in enumerate(lats):
        lat.set_gauge(syn.gauge_field(71, T, L, L, L, world, r))
    Vh = lats[0].Vh
    for ieo in (0, 1):
        kg = g.new_field(); kg[:g.Vh] = syn.spinor_field_eo(72, 1 - ieo, Tg, L, L, L)
        ref = g.new_field()
        g.Hopping_Matrix(ieo, ref, kg)
        ks = [lat.field(syn.spinor_field_eo(72, 1 - ieo, T, L, L, L, world, r)) for r, lat in enumerate(lats)]
        ls = [lat.field() for lat in lats]
        for _ in range(2):
            multi_Hopping_Matrix(lats, ieo, ls, ks)
        for r in range(world):
            assert rel_err(ls[r].download(), ref[r * Vh:(r + 1) * Vh]) < TOL, (ieo, r)
        for f in ks + ls:
            f.free()
    for lat in lats:
        lat.close()


def test_fused_cg_on_a_t8_rank_of_32_cubed():
    """One rank's share of configs[3] on 8 GPUs (8 x 32^3) with the split-phase path and the reductions of the fused CG iteration
    (cg_enqueue_fused_qtm: boundary-kernel partials + interior partials + all-reduce) exchanging with itself, i.e. the periodic
    8 x 32^3 lattice: iteration count, residual history and solution == the unsplit solve of the same lattice."""
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, L = 8, 32
    lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
    lat.set_gauge(syn.gauge_field(73, T, L, L, L))
    N = lat.Vh
    dq, dp = lat.field(syn.spinor_field_eo(74, 1, T, L, L, L)), lat.field()
    it0, h0 = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
    ref = dp.download()
    assert it0 > 0
    for loop in (1, 2, 3):
        lat.set_loopback(loop)
        dp.zero()
        it, h = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
        assert abs(it - it0) <= 1, (loop, it, it0)
        m = min(len(h), len(h0)) - 1
        assert np.allclose(h[:m], h0[:m], rtol=1e-6), loop
        assert rel_err(dp.download(), ref) < 1e-9, loop
    lat.close()


@pytest.mark.parametrize("tag", ["4x4", "8x8"])
def test_mixed_cg_her_restart_points_against_reference_restatement(tag):
    """Same RANLUX-seeded gauge field and source as the fixture run (oracle/make_golden.py mixed): iteration count as
    mixed_cg_her.c:186 returns it, number of outer iterations and the inner iteration count j of each (:141 restart rule, :152).
    fp32 rounding differs (full spinors vs half spinors, reduction order), so a restart may move by an iteration."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    s = json.load(open(os.path.join(GOLD, "ref_mixed_scalars_%s.json" % tag)))
    T, L = s["T"], s["L"]
    if tag == "4x4":
        f = np.load(os.path.join(GOLD, "ref_fields_4x4.npz"))
        gauge, src = np.ascontiguousarray(f["gauge"]), np.ascontiguousarray(f["in"])
        sol = np.load(os.path.join(GOLD, "ref_mixed_fields_4x4.npz"))
    else:   # the 8^4 inputs are not committed (19 MB): regenerate them with the oracle's RANLUX restatement when the reference library is absent
        from oracle import refbind
        if not refbind.ref_available(hs=True):
            pytest.skip("8^4 inputs come from the reference's RANLUX start-up (oracle/_ref)")
        import subprocess, sys, tempfile
        out = tempfile.NamedTemporaryFile(suffix=".npz", delete=False).name
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from oracle.refbind import RefLattice; r = RefLattice(8, 8, 8, 8, kappa=0.125, mu=0.01, nfields=8, hs=True); "
                "r.random_fields(123456); np.savez(%r, gauge=r.gauge().copy(), src=r.spinor(0, r.V // 2).copy())" % (ROOT, out))
        subprocess.check_call([sys.executable, "-c", code])     # one reference lattice per process: its state lives in C globals
        z = np.load(out)
        gauge, src = np.ascontiguousarray(z["gauge"]), np.ascontiguousarray(z["src"])
        os.unlink(out)
        sol = None
    lat = Lattice(T, L, L, L, kappa=s["kappa"], mu=s["mu"])
    orc = Oracle(T, L, L, L, kappa=s["kappa"], mu=s["mu"], threads=4)
    lat.set_gauge(gauge); orc.set_gauge(gauge)
    N = lat.Vh
    dq, dp = lat.field(src), lat.field()
    for run in s["runs"]:
        it, outer = lat.mixed_cg_her(dp, dq, 2000, s["eps_sq"], s["rel_prec"], N, innereps=run["innereps"], max_inner_it=run["max_inner_it"])
        js = lat.mixed_cg_restarts()
        assert it > 0 and len(js) == outer and it == sum(js) + 2 * (outer - 1)   # :152 iter += j, :196 iter++ after all but the last, :186 return iter + i
        assert abs(outer - len(run["inner_iters"])) <= 1, (run, js)
        assert abs(it - run["iters"]) <= max(3, run["iters"] // 8), (run, it, js)
        for a, b in zip(js[:-1], run["inner_iters"][:-1]):                      # every restart but the last (which depends on the exit taken)
            assert abs(a - b) <= 1, (run, js)
        if run["max_inner_it"] < 100:
            assert max(js) <= run["max_inner_it"]
        x = dp.download()
        full = orc.new_field(); full[:N] = x
        chk = orc.new_field(); orc.op("Qtm_pm_psi", chk, full)
        assert ((chk[:N] - src) ** 2).sum() / (src ** 2).sum() <= s["eps_sq"]
        if sol is not None:
            assert rel_err(x, sol["solution_innereps_%g_maxinner_%d" % (run["innereps"], run["max_inner_it"])]) < 1e-8
    lat.close()


def test_multi_hopping_back_to_back_with_a_delayed_rank():
    """tmhip_multi_hopping_matrix called back to back on different inputs while one rank's main stream is kept busy: the new faces
    must not land in recv_up / recv_dn while that rank's previous boundary kernel still reads them."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    from tmlqcd_amd.hip import multi_Hopping_Matrix
    T, L, world = 4, 16, 3
    Tg = T * world
    g = Oracle(Tg, L, L, L, kappa=0.125, threads=8)
    g.set_gauge(syn.gauge_field(81, Tg, L, L, L))
    lats = [Lattice(T, L, L, L, kappa=0.125, nproc_t=world, proc_t=r) for r in range(world)]
    for r, lat in enumerate(lats):
        lat.set_gauge(syn.gauge_field(81, T, L, L, L, world, r))
    Vh = lats[0].Vh
    refs, kss = [], []
    for seed in (82, 83, 84):
        kg = g.new_field(); kg[:g.Vh] = syn.spinor_field_eo(seed, 1, Tg, L, L, L)
        ref = g.new_field(); g.Hopping_Matrix(0, ref, kg)
        refs.append(ref[:g.Vh].copy())
        kss.append([lat.field(syn.spinor_field_eo(seed, 1, T, L, L, L, world, r)) for r, lat in enumerate(lats)])
    lss = [[lat.field() for lat in lats] for _ in range(3)]
    busy_a, busy_b = lats[1].field(syn.spinor_field_eo(85, 1, T, L, L, L, world, 1)), lats[1].field()
    for rep in range(3):
        for i in range(3):
            if i == 1:                                   # hold rank 1 back: a queue of work on its main stream in front of its pack kernel
                for _ in range(40):
                    lats[1].Hopping_Matrix_nocom(0, busy_b, busy_a)
            multi_Hopping_Matrix(lats, 0, lss[i], kss[i])
        for i in range(3):
            for r in range(world):
                assert rel_err(lss[i][r].download(), refs[i][r * Vh:(r + 1) * Vh]) < TOL, (rep, i, r)
    for lat in lats:
        lat.close()
