"""GPU parity through the C-ABI: linalg, site-diagonal ops, e/o compositions, D_psi and cg_her
against the CPU oracle on the same seeded inputs, plus the reference-generated golden fixtures."""
import json
import os

import numpy as np
import pytest

from tests.util import TOL, random_gauge, random_spinor, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def setup():
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = 8, 6, 4, 12          # ragged extents: no power of two, LZ/2 odd
    kappa, mu, theta = 0.131, 0.017, (1.0, 0.5, -0.25, 0.125)
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(41, orc.VPR)
    orc.set_gauge(g)
    lat.set_gauge(g)
    yield orc, lat
    lat.close()


def test_reductions(setup):
    orc, lat = setup
    N = orc.Vh
    a, b = random_spinor(1, N), random_spinor(2, N)
    da, db = lat.field(a), lat.field(b)
    for got, ref in ((lat.square_norm(da, N), orc.square_norm(a, N)),
                     (lat.scalar_prod_r(da, db, N), orc.scalar_prod_r(a, b, N))):
        assert abs(got - ref) <= TOL * abs(orc.square_norm(a, N))
    ra = a.copy()
    nref = orc.assign_mul_add_r_and_square(ra, -0.7, b, N)
    ngot = lat.assign_mul_add_r_and_square(da, -0.7, db, N)
    assert abs(ngot - nref) <= TOL * nref and rel_err(da.download(), ra) < TOL
    da.free(); db.free()


def test_reduction_is_reproducible_and_accurate(setup):
    """Fixed-order tree: bitwise identical run to run; vs an exactly rounded sum (math.fsum) < 1e-14."""
    import math
    orc, lat = setup
    N = orc.Vh
    a = random_spinor(7, N) * np.logspace(-3, 3, N)[:, None, None, None]
    da = lat.field(a)
    v = [lat.square_norm(da, N) for _ in range(3)]
    assert v[0] == v[1] == v[2]
    exact = math.fsum((a.astype(np.float64) ** 2).ravel().tolist())
    assert abs(v[0] - exact) <= 1e-14 * exact
    da.free()


def test_streaming_updates(setup):
    orc, lat = setup
    N = orc.Vh
    a, b = random_spinor(3, N), random_spinor(4, N)
    da, db, dc = lat.field(a), lat.field(b), lat.field()
    ra = a.copy(); orc.assign_add_mul_r(ra, b, 0.756, N); lat.assign_add_mul_r(da, db, 0.756, N)
    assert rel_err(da.download(), ra) < TOL
    orc.assign_mul_add_r(ra, -1.3, b, N); lat.assign_mul_add_r(da, -1.3, db, N)
    assert rel_err(da.download(), ra) < TOL
    rc = np.zeros_like(a); orc.diff(rc, ra, b, N); lat.diff(dc, da, db, N)
    assert rel_err(dc.download(), rc) < TOL
    lat.assign(dc, db, N)
    assert np.array_equal(dc.download(), b)
    mine = np.full_like(b, 7.0)                                   # download into the caller's own array
    assert dc.download(out=mine) is mine and np.array_equal(mine, b)
    with pytest.raises(Exception):
        dc.download(out=np.zeros((N, 4, 3), dtype=np.float64))
    da.upload(np.ascontiguousarray(ra))                           # identical inputs from here on
    orc.add(rc, ra, b, N); lat.add(dc, da, db, N)                 # linalg/add.c, linalg/mul_r.c: one rounding each, so bit for bit
    assert np.array_equal(dc.download(), rc)
    orc.mul_r(rc, -0.73, b, N); lat.mul_r(dc, -0.73, db, N)
    assert np.array_equal(dc.download(), rc)
    lat.mul_r(dc, 2.0, dc, N)                                     # in place, as operator.c normalises a propagator
    assert np.array_equal(dc.download(), 2.0 * rc)
    for f in (da, db, dc):
        f.free()


@pytest.mark.parametrize("sign", [+1.0, -1.0])
def test_site_diagonal_ops_and_aliasing(setup, sign):
    orc, lat = setup
    N = orc.Vh
    k, j = random_spinor(5, N), random_spinor(6, N)
    dk, dj, dl = lat.field(k), lat.field(j), lat.field()
    ref = np.zeros_like(k)
    orc.assign_mul_one_pm_imu_inv(ref, k, sign, N); lat.assign_mul_one_pm_imu_inv(dl, dk, sign, N)
    assert rel_err(dl.download(), ref) < TOL
    orc.assign_mul_one_pm_imu(ref, k, sign, N); lat.assign_mul_one_pm_imu(dl, dk, sign, N)
    assert rel_err(dl.download(), ref) < TOL
    orc.mul_one_pm_imu_sub_mul(ref, k, j, sign, N); lat.mul_one_pm_imu_sub_mul(dl, dk, dj, sign, N)
    assert rel_err(dl.download(), ref) < TOL
    orc.mul_one_pm_imu_sub_mul_gamma5(ref, k, j, sign); lat.mul_one_pm_imu_sub_mul_gamma5(dl, dk, dj, sign)
    assert rel_err(dl.download(), ref) < TOL
    # in place, as tm_operators.c:371 (l == j) and mul_one_pm_imu_inv (l == k) use them
    rj = j.copy(); orc.mul_one_pm_imu_sub_mul_gamma5(rj, k, rj, sign); lat.mul_one_pm_imu_sub_mul_gamma5(dj, dk, dj, sign)
    assert rel_err(dj.download(), rj) < TOL
    rk = k.copy(); orc.mul_one_pm_imu_inv(rk, sign, N); lat.mul_one_pm_imu_inv(dk, sign, N)
    assert rel_err(dk.download(), rk) < TOL
    dk.upload(rk)
    orc.gamma5(ref, rk, N); lat.gamma5(dl, dk, N)
    assert np.array_equal(dl.download(), ref)              # sign flips only: exact
    for f in (dk, dj, dl):
        f.free()


@pytest.mark.parametrize("name", ["Qtm_pm_psi", "Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi",
                                  "Qtm_plus_sym_psi", "Qtm_minus_sym_psi", "Mtm_plus_sym_psi", "Mtm_minus_sym_psi",
                                  "Mtm_plus_sym_dagg_psi", "Qtm_pm_sym_psi"])
def test_eo_operators(setup, name):
    orc, lat = setup
    N = orc.Vh
    k = random_spinor(8, N)
    ref = orc.new_field()
    orc.op(name, ref, k.copy())
    dk, dl = lat.field(k), lat.field()
    lat.op(name, dl, dk)
    assert rel_err(dl.download(), ref[:N]) < TOL
    assert np.array_equal(dk.download(), k)          # input untouched
    dk.free(); dl.free()


def test_qtm_minus_in_place_like_invert_eo(setup):
    """invert_eo.c:270 calls Qtm_minus_psi(Odd_new, Odd_new)."""
    orc, lat = setup
    N = orc.Vh
    k = random_spinor(9, N)
    ref = orc.new_field(); ref[:N] = k
    orc.op("Qtm_minus_psi", ref, ref)
    dk = lat.field(k)
    lat.op("Qtm_minus_psi", dk, dk)
    assert rel_err(dk.download(), ref[:N]) < TOL
    dk.free()


def test_M_full_and_D_psi(setup):
    orc, lat = setup
    N, V = orc.Vh, orc.V
    e, o = random_spinor(10, N), random_spinor(11, N)
    ren, ron = orc.new_field(), orc.new_field()
    orc.M_full(ren, ron, e, o)
    de, do, den, don = lat.field(e), lat.field(o), lat.field(), lat.field()
    lat.M_full(den, don, de, do)
    assert rel_err(den.download(), ren[:N]) < TOL and rel_err(don.download(), ron[:N]) < TOL
    lex = random_spinor(12, V)
    P = np.zeros_like(lex)
    orc.D_psi(P, lex)
    dQ = lat.full_field(lex)
    dP = lat.full_field()
    lat.D_psi(dP, dQ)
    assert rel_err(dP.download(), P) < TOL
    assert np.array_equal(dQ.download(), lex)        # lexicographic <-> e/o round trip is exact
    from tmlqcd_amd.hip import TmHipError
    with pytest.raises(TmHipError):                  # D_psi_body.c:267-272: P == Q is an error
        lat.D_psi(dQ, dQ)
    with pytest.raises(TmHipError):                  # stencil: l != k
        lat.Hopping_Matrix(0, de, de)
    for f in (de, do, den, don, dQ, dP):
        f.free()


def test_linearity_and_hermiticity_at_scale():
    """Size-independent properties at a BASELINE size (16^4): H(a x + y) = a H x + H y;
    <y, Q+Q- x> = <Q+Q- y, x> (Qtm_pm_psi is hermitian, which is what cg_her relies on)."""
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T = L = 16
    lat = Lattice(T, L, L, L, kappa=0.137, mu=0.02)
    lat.set_gauge(syn.gauge_field(5, T, L, L, L))
    N = lat.Vh
    x, y = lat.field(syn.spinor_field_eo(6, 1, T, L, L, L)), lat.field(syn.spinor_field_eo(7, 1, T, L, L, L))
    hx, hy, z, hz = lat.field(), lat.field(), lat.field(), lat.field()
    lat.Hopping_Matrix(0, hx, x); lat.Hopping_Matrix(0, hy, y)
    lat.assign(z, y, N); lat.assign_add_mul_r(z, x, 0.37, N)          # z = 0.37 x + y
    lat.Hopping_Matrix(0, hz, z)
    lat.assign_add_mul_r(hy, hx, 0.37, N)                              # hy = 0.37 Hx + Hy
    lat.diff(hz, hz, hy, N)
    assert lat.square_norm(hz, N) <= (1e-13) ** 2 * lat.square_norm(hy, N)
    qx, qy = lat.field(), lat.field()
    lat.Qtm_pm_psi(qx, x); lat.Qtm_pm_psi(qy, y)
    a, b = lat.scalar_prod_r(y, qx, N), lat.scalar_prod_r(qy, x, N)
    assert abs(a - b) <= 1e-12 * abs(a)
    assert lat.scalar_prod_r(x, qx, N) > 0                             # positive
    lat.close()


@pytest.mark.parametrize("cg_sync,batch", [(0, 4), (0, 1), (0, 7), (1, 1)])
def test_cg_her_matches_oracle(setup, cg_sync, batch):
    """cg_sync=1: reference loop with host round trips; cg_sync=0: scalars and stopping test on the device,
    host polls the done flag every `batch` iterations -- iteration count and solution must not depend on it."""
    orc, lat = setup
    lat.set_option("cg_sync", cg_sync); lat.set_option("cg_batch", batch)
    N = orc.Vh
    q = random_spinor(13, N)
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, q.copy(), 500, 1e-20, 1, N)
    dq, dp = lat.field(q), lat.field()
    it, hist = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
    assert it > 0 and abs(it - it_ref) <= 1                    # within +-1 iteration
    m = min(len(hist), len(hist_ref)) - 1
    assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6)      # residual histories track each other
    sol = dp.download()
    assert rel_err(sol, P[:N]) < 1e-9
    # true residual in fp64 on the CPU (operator.c:358,379-384 "reached_prec")
    chk = orc.new_field(); full = orc.new_field(); full[:N] = sol
    orc.op("Qtm_pm_psi", chk, full)
    res = ((chk[:N] - q) ** 2).sum() / (q ** 2).sum()
    assert res <= 4e-20
    assert len(hist) == it
    lat.set_option("cg_sync", 0); lat.set_option("cg_batch", 4)
    dq.free(); dp.free()


@pytest.mark.parametrize("fused", [2, 1, 0])
def test_cg_fused_scalar_product_path(fused):
    """V/2 % 256 == 0 (8^4).  2 (default): pro = |Q_- p|^2 from the second stencil, r -= alpha A p and |r|^2 in the
    epilogue of the fourth, one pass for (P, p); 1: scalar_prod_r fused into the last stencil only; 0: separate linalg
    kernels.  Same iterations, same residual history, same solution as the oracle for all three."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T = L = 8
    orc = Oracle(T, L, L, L, kappa=0.13, mu=0.015, threads=8)
    lat = Lattice(T, L, L, L, kappa=0.13, mu=0.015)
    g = syn.gauge_field(21, T, L, L, L)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    q = syn.spinor_field_eo(22, 1, T, L, L, L)
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, q.copy(), 500, 1e-20, 1, N)
    lat.set_option("cg_fused_dot", fused)
    dq, dp = lat.field(q), lat.field()
    it, hist = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
    assert abs(it - it_ref) <= 1
    m = min(len(hist), len(hist_ref)) - 1
    assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6)
    assert rel_err(dp.download(), P[:N]) < 1e-9
    # a non-zero initial guess, an iteration cap that ends mid-batch and a second solve on the same context
    x0 = 0.5 * P[:N] + 0.01 * syn.spinor_field_eo(23, 1, T, L, L, L)
    Pr = orc.new_field(); Pr[:N] = x0
    it_ref2, _ = orc.cg_her(Pr, q.copy(), 500, 1e-20, 1, N)
    dp.upload(np.ascontiguousarray(x0))
    it2, _ = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
    assert abs(it2 - it_ref2) <= 1 and rel_err(dp.download(), Pr[:N]) < 1e-9
    dp.zero()
    it3, hist3 = lat.cg_her(dp, dq, 7, 1e-30, 1, N)
    Pc = orc.new_field(); it_c, hist_c = orc.cg_her(Pc, q.copy(), 7, 1e-30, 1, N)
    assert it3 == -1 and it_c == -1 and len(hist3) == 7
    assert rel_err(dp.download(), Pc[:N]) < 1e-9          # P after exactly 7 updates, no extra / missing alpha p
    # the same solve on the split-phase path (T-split rank rehearsed on one GPU): with cg_fused_dot = 2 the reductions are
    # spread over the stencil kernel (all sites but the two boundary slices) and the exterior kernel (those two)
    for mode, ss in ((1, 0), (2, 0), (1, 1), (2, 1), (3, 8), (3, 16)):      # ss = 1: HIP events instead of flags; 8 / 16: direct carrier, one kernel / stencil + exterior kernel
        lat.set_option("split_sync", ss & 1); lat.set_option("direct_form", 1 if ss & 8 else (0 if ss & 16 else -1))
        lat.set_loopback(mode)
        dp.zero()
        it4, hist4 = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
        lat.set_loopback(0); lat.set_option("split_sync", 0); lat.set_option("direct_form", -1)
        assert abs(it4 - it_ref) <= 1 and rel_err(dp.download(), P[:N]) < 1e-9, (mode, ss)
        m4 = min(len(hist4), len(hist_ref)) - 1
        assert np.allclose(hist4[:m4], hist_ref[:m4], rtol=1e-6)
    lat.close()


def test_cg_fused_split_path_wide_short_local_lattice():
    """A T-split rank whose faces are a large part of its volume (4 x 16^3: half the sites are face sites): the exterior kernel
    owns half of the reduction.  Must reproduce the unsplit solve."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, L = 4, 16
    lat = Lattice(T, L, L, L, kappa=0.13, mu=0.015)
    orc = Oracle(T, L, L, L, kappa=0.13, mu=0.015, threads=8)
    g = syn.gauge_field(31, T, L, L, L)
    lat.set_gauge(g); orc.set_gauge(g)
    N = lat.Vh
    q = syn.spinor_field_eo(32, 1, T, L, L, L)
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, q.copy(), 500, 1e-20, 1, N)
    dq, dp = lat.field(q), lat.field()
    for loop, ss in ((0, 0), (1, 0), (2, 0), (1, 1), (3, 8), (3, 16)):      # ss 8 / 16: direct carrier, one kernel / two kernels
        lat.set_option("split_sync", ss & 1); lat.set_option("direct_form", 1 if ss & 8 else (0 if ss & 16 else -1))
        lat.set_loopback(loop)
        dp.zero()
        it, hist = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
        assert abs(it - it_ref) <= 1, (loop, ss, it, it_ref)
        m = min(len(hist), len(hist_ref)) - 1
        assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6), (loop, ss)
        assert rel_err(dp.download(), P[:N]) < 1e-9, (loop, ss)
    lat.close()


def test_cg_fused_with_one_communicator_for_faces_and_reductions():
    """The fallback of an RCCL without ncclCommSplit, forced ("comm_split" 0): the scalar all-reduces of the fused CG iteration run
    on the SAME communicator as the face exchange, from the other stream.  Correct because the exchange of a stencil has completed
    before anything enqueued behind that stencil on the main stream starts (launch_split, hopping_split.inc).  One-rank RCCL
    loopback on 8 x 16^3 == the unsplit solve; the library says which form it runs."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, L = 8, 16
    orc = Oracle(T, L, L, L, kappa=0.13, mu=0.015, threads=8)
    g = syn.gauge_field(41, T, L, L, L)
    orc.set_gauge(g)
    q = syn.spinor_field_eo(42, 1, T, L, L, L)
    N = orc.Vh
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, q.copy(), 500, 1e-20, 1, N)
    for split in (0, 1):
        lat = Lattice(T, L, L, L, kappa=0.13, mu=0.015)
        lat.set_gauge(g)
        assert lat.comm_is_split() is None
        lat.set_option("comm_split", split)
        lat.set_loopback(2)
        assert lat.comm_is_split() is bool(split) and lat.comm_count() == (1, 1)
        dq, dp = lat.field(q), lat.field()
        it, hist = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
        assert abs(it - it_ref) <= 1, (split, it, it_ref)
        m = min(len(hist), len(hist_ref)) - 1
        assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6), split
        assert rel_err(dp.download(), P[:N]) < 1e-9, split
        nrm = lat.square_norm(dp, N, 1)                       # a plain global reduction on that communicator as well
        assert abs(nrm - orc.square_norm(P, N)) <= 1e-12 * nrm
        lat.close()


@pytest.mark.parametrize("dims", [(6, 4, 4, 4), (4, 4, 4, 8), (12, 4, 4, 4), (2, 4, 6, 2)])
def test_cg_fused_iteration_on_small_block_counts(dims):
    """V/2 = 192, 256, 384: whole 64-thread blocks but not (always) whole 256-thread blocks -- the fused iteration runs with the
    64-thread instances of the reducing epilogues (3, 4, 6 blocks); V/2 = 48 has no whole block and takes the plain kernels.
    All against the oracle's cg_her, twisted boundary conditions in every direction."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = dims
    theta = (1.0, 0.3, -0.2, 0.5)
    orc = Oracle(T, LX, LY, LZ, kappa=0.135, mu=0.03, theta=theta, threads=4)
    lat = Lattice(T, LX, LY, LZ, kappa=0.135, mu=0.03, theta=theta)
    g = random_gauge(sum(dims) + 1, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    q = random_spinor(41, N)
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, q.copy(), 500, 1e-22, 1, N)
    for fused in (2, 0):
        lat.set_option("cg_fused_dot", fused)
        dq, dp = lat.field(q), lat.field()
        it, hist = lat.cg_her(dp, dq, 500, 1e-22, 1, N)
        assert abs(it - it_ref) <= 1, (fused, it, it_ref)
        m = min(len(hist), len(hist_ref)) - 1
        assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6), fused
        assert rel_err(dp.download(), P[:N]) < 1e-9, fused
    lat.close()


@pytest.mark.parametrize("block", [256, 64])
def test_cg_fused_iteration_with_both_block_sizes(block):
    """The automatic block size picks 64 threads on every test-sized lattice; the 256-thread instances of the reducing epilogues
    are what runs at 32^4.  Both, forced, on 8^4 (unsplit and on the split path) against the oracle."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T = L = 8
    orc = Oracle(T, L, L, L, kappa=0.13, mu=0.015, threads=8)
    lat = Lattice(T, L, L, L, kappa=0.13, mu=0.015)
    g = syn.gauge_field(21, T, L, L, L)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    q = syn.spinor_field_eo(22, 1, T, L, L, L)
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, q.copy(), 500, 1e-20, 1, N)
    lat.set_option("block", block)
    dq, dp = lat.field(q), lat.field()
    for loop in (0, 1, 3):
        lat.set_loopback(loop)
        dp.zero()
        it, hist = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
        assert abs(it - it_ref) <= 1, (block, loop)
        m = min(len(hist), len(hist_ref)) - 1
        assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6) and rel_err(dp.download(), P[:N]) < 1e-9, (block, loop)
    it32, outer = lat.mixed_cg_her(dp, dq, 500, 1e-20, 1, N)     # fp32 inner loops: the same epilogues in float
    assert rel_err(dp.download(), P[:N]) < 1e-8, block
    lat.close()


def test_cg_not_converged_returns_minus_one(setup):
    orc, lat = setup
    N = orc.Vh
    dq, dp = lat.field(random_spinor(14, N)), lat.field()
    it, _ = lat.cg_her(dp, dq, 3, 1e-30, 1, N)
    assert it == -1                                            # cg_her.c:140-141
    dq.free(); dp.free()


def test_golden_fixture_from_reference_on_gpu():
    """tests/golden/ref_fields_4x4.npz: inputs and outputs produced by the reference object code."""
    from tmlqcd_amd import Lattice
    f = np.load(os.path.join(GOLD, "ref_fields_4x4.npz"))
    s = json.load(open(os.path.join(GOLD, "ref_scalars_4x4.json")))
    lat = Lattice(4, 4, 4, 4, kappa=s["kappa"], mu=s["mu"])
    lat.set_gauge(np.ascontiguousarray(f["gauge"]))
    N = lat.Vh
    d0, d1, d2 = lat.field(np.ascontiguousarray(f["in"])), lat.field(), lat.field()
    lat.Hopping_Matrix(0, d1, d0); lat.Hopping_Matrix(1, d2, d1)
    assert rel_err(d1.download(), f["Heo"]) < TOL and rel_err(d2.download(), f["HoeHeo"]) < TOL
    assert abs(lat.square_norm(d2, N) - s["norm_HoeHeo"]) <= TOL * s["norm_HoeHeo"]
    for name in ("Qtm_pm_psi", "Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi"):
        lat.op(name, d2, d0)
        assert rel_err(d2.download(), f[name]) < TOL, name
    g = np.load(os.path.join(GOLD, "ref_sym_fields_4x4.npz"))
    for name in g.files:                      # symmetric preconditioning family, tm_operators.c:186-364
        lat.op(name, d2, d0)
        assert rel_err(d2.download(), g[name]) < TOL, name
    c = complex(*s["cfactor"])
    lat.tm_times_Hopping_Matrix(1, d2, d1, c)
    assert rel_err(d2.download(), f["tm_times_OE_of_Heo"]) < TOL
    lat.tm_sub_Hopping_Matrix(1, d2, d0, d1, c)
    assert rel_err(d2.download(), f["tm_sub_OE_p_in_k_Heo"]) < TOL
    dQ, dP = lat.full_field(np.ascontiguousarray(f["D_psi_in_lexic"])), lat.full_field()
    lat.D_psi(dP, dQ)
    assert rel_err(dP.download(), f["D_psi_out_lexic"]) < TOL
    dp = lat.field()
    it, _ = lat.cg_her(dp, d0, 1000, 1e-20, 1, N)
    assert abs(it - s["cg_iters"]) <= 1
    assert rel_err(dp.download(), f["cg_solution"]) < 1e-9
    lat.close()


@pytest.mark.parametrize("T,world", [(2, 2), (4, 2), (2, 3), (2, 4)])
def test_t_split_two_contexts_on_one_gpu(T, world):
    """nproc_t = 2, 3, 4 geometry on real hardware: as many contexts of this process hold the slabs (halo gauge links, global
    parity offset, face pack -> peer copy -> boundary kernels) == the unsplit lattice.  From three ranks on, the up and the
    down neighbour are different ranks."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    from tmlqcd_amd.hip import multi_Hopping_Matrix
    L = 6
    Tg = T * world
    kappa, theta = 0.13, (1.0, 0.0, 0.0, 0.5)
    g = Oracle(Tg, L, L, L, kappa=kappa, theta=theta, threads=4)
    g.set_gauge(syn.gauge_field(3, Tg, L, L, L))
    lats = [Lattice(T, L, L, L, kappa=kappa, theta=theta, nproc_t=world, proc_t=r) for r in range(world)]
    for r, lat in enumerate(lats):
        lat.set_gauge(syn.gauge_field(3, T, L, L, L, world, r))
    Vh = lats[0].Vh
    for ieo in (0, 1):
        kg = g.new_field(); kg[:g.Vh] = syn.spinor_field_eo(4, 1 - ieo, Tg, L, L, L)
        ref = g.new_field()
        g.Hopping_Matrix(ieo, ref, kg)
        ks = [lat.field(syn.spinor_field_eo(4, 1 - ieo, T, L, L, L, world, r)) for r, lat in enumerate(lats)]
        ls = [lat.field() for lat in lats]
        for _ in range(2):   # twice: the second call exercises the buffer re-use ordering
            multi_Hopping_Matrix(lats, ieo, ls, ks)
        for r in range(world):
            assert rel_err(ls[r].download(), ref[r * Vh:(r + 1) * Vh]) < TOL, (ieo, r)
    for lat in lats:
        lat.close()


def test_linalg_empty_and_out_of_range_site_counts(setup):
    """N = 0 is a legal empty loop in the reference's linalg (returns 0 / leaves the fields alone); N outside [0, V/2] must
    fail loudly instead of running past the device arrays."""
    from tmlqcd_amd.hip import TmHipError
    orc, lat = setup
    N = orc.Vh
    a, b = random_spinor(401, N), random_spinor(402, N)
    da, db = lat.field(a), lat.field(b)
    assert lat.square_norm(da, 0) == 0.0 and lat.scalar_prod_r(da, db, 0) == 0.0
    lat.assign_add_mul_r(da, db, 0.5, 0); lat.diff(da, da, db, 0); lat.assign(da, db, 0); lat.add(da, da, db, 0); lat.mul_r(da, 3.0, db, 0)
    assert np.array_equal(da.download(), a)
    half = N // 2 + 3                                        # ragged prefix: only the first `half` sites take part
    assert abs(lat.square_norm(da, half) - (a[:half] ** 2).sum()) <= 1e-13 * (a[:half] ** 2).sum()
    lat.assign_add_mul_r(da, db, 0.25, half)
    out = da.download()
    assert rel_err(out[:half], a[:half] + 0.25 * b[:half]) < TOL and np.array_equal(out[half:], a[half:])
    for bad in (-1, N + 1):
        with pytest.raises(TmHipError):
            lat.square_norm(da, bad)
        with pytest.raises(TmHipError):
            lat.assign_add_mul_r(da, db, 1.0, bad)
    da.free(); db.free()


def test_direct_carrier_set_up_over_a_one_rank_rccl_communicator():
    """tmhip_comm_init_ipc's collective set-up as it runs between GPUs -- the gather of the IPC cards and the all-or-none sum over RCCL
    (ncclAllGather / ncclAllReduce on the reduction communicator) -- rehearsed behind the one-rank communicator of loopback 2, where
    the rank becomes its own neighbour: faces as direct stores, the scalar sums as direct sums (one wave, np = 1), cg_her with the
    reduction fused into one launch.  Everything must equal the unsplit lattice."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, L = 8, 16
    lat = Lattice(T, L, L, L, kappa=0.13, mu=0.015)
    orc = Oracle(T, L, L, L, kappa=0.13, mu=0.015, threads=8)
    g = syn.gauge_field(41, T, L, L, L)
    lat.set_gauge(g); orc.set_gauge(g)
    N = lat.Vh
    q = syn.spinor_field_eo(42, 1, T, L, L, L)
    lat.set_loopback(2)
    lat.comm_init_ipc()
    assert lat.comm_faces_direct() == (True, 1) and lat.comm_sums_direct() is True and lat.comm_count() == (1, 1)
    dq, dl, dp = lat.field(q), lat.field(), lat.field()
    ref = orc.new_field()
    for ieo in (0, 1):
        orc.Hopping_Matrix(ieo, ref, q); lat.Hopping_Matrix(ieo, dl, dq)
        assert rel_err(dl.download(), ref[:N]) < TOL
    orc.op("Qtm_pm_psi", ref, q.copy()); lat.Qtm_pm_psi(dl, dq)
    assert rel_err(dl.download(), ref[:N]) < TOL
    n_ref = orc.square_norm(ref, N)
    assert abs(lat.square_norm(dl, N, 1) - n_ref) <= 1e-13 * n_ref            # parallel = 1: through the direct sum
    P = orc.new_field()
    it_ref, hist_ref = orc.cg_her(P, q.copy(), 500, 1e-20, 1, N)
    it, hist = lat.cg_her(dp, dq, 500, 1e-20, 1, N)
    assert abs(it - it_ref) <= 1 and rel_err(dp.download(), P[:N]) < 1e-9
    m = min(len(hist), len(hist_ref)) - 1
    assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6)
    lat.close()
