"""GPU: clover twisted-mass operators (the invert_clover_eo path) against the reference's
fixture (sw_term / sw_invert outputs at 4^4) and against the CPU oracle on synthetic clover blocks at 8^4."""
import json
import os

import numpy as np
import pytest

from tests.util import TOL, random_clover, random_gauge, random_spinor, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_clover_fixture_from_reference():
    from tmlqcd_amd import Lattice
    f = np.load(os.path.join(GOLD, "ref_fields_4x4.npz"))
    c = np.load(os.path.join(GOLD, "ref_clover_fields_4x4.npz"))
    s = json.load(open(os.path.join(GOLD, "ref_clover_scalars_4x4.json")))
    mu = s["mu"]
    lat = Lattice(4, 4, 4, 4, kappa=s["kappa"], mu=mu)
    lat.set_gauge(np.ascontiguousarray(f["gauge"]))
    lat.set_clover(np.ascontiguousarray(c["sw"]), np.ascontiguousarray(c["sw_inv"]))
    N = lat.Vh
    k = np.ascontiguousarray(f["in"])
    dk, dl, dh = lat.field(k), lat.field(), lat.field()
    for sign, key in ((-1, "clover_inv_minus"), (+1, "clover_inv_plus")):
        dl.upload(k); lat.clover_inv(dl, sign, mu)
        assert rel_err(dl.download(), c[key]) < TOL
    lat.Hopping_Matrix(1, dh, dk)
    lat.clover_gamma5(1, dl, dk, dh, -mu)
    assert rel_err(dl.download(), c["clover_gamma5_OO_in_Hoe"]) < TOL
    lat.op("Qsw_pm_psi", dl, dk)
    assert rel_err(dl.download(), c["Qsw_pm_psi"]) < TOL
    lat.op("Msw_plus_psi", dl, dk)
    assert rel_err(dl.download(), c["Msw_plus_psi"]) < TOL
    dp = lat.field()
    it, _ = lat.cg_her(dp, dk, 1000, 1e-20, 1, N, op="Qsw_pm_psi")
    assert abs(it - s["cg_iters"]) <= 1 and rel_err(dp.download(), c["cg_solution"]) < 1e-9
    lat.close()


@pytest.fixture(scope="module")
def setup8():
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = 8, 6, 4, 12
    kappa, mu, theta = 0.13, 0.02, (1.0, 0.0, 0.3, 0.5)
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(71, orc.VPR)
    sw, swi = random_clover(72, orc, mu, scale=0.05)   # well-conditioned: O(50) CG iterations, no rounding-driven wander
    orc.set_gauge(g); lat.set_gauge(g)
    orc.set_clover(sw, swi); lat.set_clover(sw, swi)
    yield orc, lat, mu
    lat.close()


def test_clover_site_ops_and_fused_operator(setup8):
    orc, lat, mu = setup8
    N = orc.Vh
    k, j = random_spinor(1, N), random_spinor(2, N)
    dk, dj, dl = lat.field(k), lat.field(j), lat.field()
    for sign in (+1, -1):
        ref = orc.new_field(); ref[:N] = k
        orc.clover_inv(ref, sign, mu)
        dl.upload(k); lat.clover_inv(dl, sign, mu)
        assert rel_err(dl.download(), ref[:N]) < TOL
        ref2 = orc.new_field(); orc.Hopping_Matrix(0, ref2, k); orc.clover_inv(ref2, sign, mu)
        lat.H_eo_sw_inv_psi(dl, dk, 0, sign, mu)                    # stencil with the fused clover_inv epilogue
        assert rel_err(dl.download(), ref2[:N]) < TOL
    for ieo in (0, 1):
        ref = orc.new_field(); orc.clover_gamma5(ieo, ref, k, j, -mu)
        lat.clover_gamma5(ieo, dl, dk, dj, -mu)
        assert rel_err(dl.download(), ref[:N]) < TOL
        orc.clover(ieo, ref, k, j, mu); lat.clover(ieo, dl, dk, dj, mu)
        assert rel_err(dl.download(), ref[:N]) < TOL
    ref = orc.new_field(); orc.op("Qsw_pm_psi", ref, k.copy())
    lat.op("Qsw_pm_psi", dl, dk)
    assert rel_err(dl.download(), ref[:N]) < TOL
    lat.set_loopback(1)                                             # clover epilogues on the split-phase path
    lat.op("Qsw_pm_psi", dl, dk)
    lat.set_loopback(0)
    assert rel_err(dl.download(), ref[:N]) < TOL
    orc.op("Msw_plus_psi", ref, k.copy()); lat.op("Msw_plus_psi", dl, dk)
    assert rel_err(dl.download(), ref[:N]) < TOL
    # the rest of the e/o family (clovertm_operators.c:201-268) and the site operators behind Msw_full / invert_clover_eo
    for name in ("Qsw_psi", "Qsw_plus_psi", "Qsw_minus_psi", "Qsw_sq_psi", "Msw_psi", "Msw_minus_psi"):
        orc.op(name, ref, k.copy()); lat.op(name, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL, name
    # g_mu3 != 0: the odd-odd clover term twists with mu + mu3 (clovertm_operators.c:208,216,238,243,258,265), the even-even inverse
    # stays the one built for mu; cg_her on Qsw_pm_psi takes its scalar products out of the same epilogues
    orc.set_mu3(0.07); lat.set_mu3(0.07)
    for name in ("Qsw_pm_psi", "Qsw_psi", "Qsw_plus_psi", "Qsw_minus_psi", "Qsw_sq_psi", "Msw_psi", "Msw_plus_psi", "Msw_minus_psi"):
        orc.op(name, ref, k.copy()); lat.op(name, dl, dk)
        assert rel_err(dl.download(), ref[:N]) < TOL, name + " with mu3"
    orc.op("Qsw_plus_psi", ref, k.copy()); orc.set_mu3(0.0); chk = orc.new_field(); orc.op("Qsw_plus_psi", chk, k.copy()); orc.set_mu3(0.07)
    assert rel_err(ref[:N], chk[:N]) > 1e-3                            # mu3 really changes the operator
    Pm = orc.new_field(); itm, _ = orc.cg_her(Pm, j.copy(), 2000, 1e-18, 1, N, "Qsw_pm_psi")
    dpm = lat.field(); itd, _ = lat.cg_her(dpm, dj, 2000, 1e-18, 1, N, op="Qsw_pm_psi")
    assert abs(itd - itm) <= max(1, itm // 100) and rel_err(dpm.download(), Pm[:N]) < 1e-7
    itx, _ = lat.mixed_cg_her(dpm, dj, 2000, 1e-18, 1, N, op="Qsw_pm_psi")           # fp32 inner operator with the same twist
    assert rel_err(dpm.download(), Pm[:N]) < 1e-7
    orc.set_mu3(0.0); lat.set_mu3(0.0)
    ref[:N] = k; orc.op("Qsw_minus_psi", ref, ref)
    dl.upload(k); lat.op("Qsw_minus_psi", dl, dl)                      # in place, invert_clover_eo.c:128
    assert rel_err(dl.download(), ref[:N]) < TOL
    for ieo in (0, 1):
        orc.assign_mul_one_sw_pm_imu(ieo, ref, k, mu); lat.assign_mul_one_sw_pm_imu(ieo, dl, dk, mu)
        assert rel_err(dl.download(), ref[:N]) < TOL
    orc.assign_mul_one_sw_pm_imu_inv(0, ref, k, mu); lat.assign_mul_one_sw_pm_imu_inv(0, dl, dk, mu)
    assert rel_err(dl.download(), ref[:N]) < TOL
    ro = orc.new_field(); do = lat.field()
    orc.Msw_full(ref, ro, k, j); lat.Msw_full(dl, do, dk, dj)
    assert rel_err(dl.download(), ref[:N]) < TOL and rel_err(do.download(), ro[:N]) < TOL
    do.free()
    for f in (dk, dj, dl):
        f.free()


def test_clover_cg_and_mixed_cg(setup8):
    """BASELINE configs[4] in miniature: clover twisted mass, fp64 CG and mixed fp32/fp64 CG with fp64 residual restart."""
    orc, lat, mu = setup8
    N = orc.Vh
    q = random_spinor(3, N)
    P = orc.new_field(); it_ref, _ = orc.cg_her(P, q.copy(), 2000, 1e-20, 1, N, "Qsw_pm_psi")
    dq, dp = lat.field(q), lat.field()
    it, _ = lat.cg_her(dp, dq, 2000, 1e-20, 1, N, op="Qsw_pm_psi")
    assert abs(it - it_ref) <= max(1, it_ref // 100) and rel_err(dp.download(), P[:N]) < 1e-9
    # fp32 operator vs fp64 oracle
    k32 = q.astype(np.float32)
    d32, l32 = lat.field32(k32), lat.field32()
    ref = orc.new_field(); orc.op("Qsw_pm_psi", ref, k32.astype(np.float64))
    lat.Qsw_pm_psi_32(l32, d32)
    assert rel_err(l32.download().astype(np.float64), ref[:N]) < 1e-5
    itm, outer = lat.mixed_cg_her(dp, dq, 5000, 1e-20, 1, N, op="Qsw_pm_psi")
    assert itm > 0 and outer >= 2
    sol = dp.download()
    full = orc.new_field(); full[:N] = sol
    chk = orc.new_field(); orc.op("Qsw_pm_psi", chk, full)
    assert ((chk[:N] - q) ** 2).sum() / (q ** 2).sum() <= 1e-20
    assert rel_err(sol, P[:N]) < 1e-8
    # reliable-update variant (solver/rg_mixed_cg_her.c:180) with f32 = Qsw_pm_psi_32
    itr, (n_out, n_sp, n_dp) = lat.rg_mixed_cg_her(dp, dq, 5000, 1e-20, 1, N, delta=0.1, op="Qsw_pm_psi")
    assert itr > 0 and itr == n_out + n_sp + n_dp and n_out >= 2
    assert rel_err(dp.download(), P[:N]) < 1e-8
    for f in (dq, dp, d32, l32):
        f.free()


# ------------------------------------------------------------------ sw_term / sw_invert computed on the device
def test_device_sw_term_and_sw_invert_against_reference_fixture():
    """operator/clover_term.c:88 + operator/clover_invert.c:170 on the GPU from the gauge field of the 4^4 fixture ==
    the arrays the reference's own sw_term / sw_invert(EE, mu) produced (oracle/make_golden.py clover)."""
    from tmlqcd_amd import Lattice
    f = np.load(os.path.join(GOLD, "ref_fields_4x4.npz"))
    c = np.load(os.path.join(GOLD, "ref_clover_fields_4x4.npz"))
    s = json.load(open(os.path.join(GOLD, "ref_clover_scalars_4x4.json")))
    lat = Lattice(4, 4, 4, 4, kappa=s["kappa"], mu=s["mu"])
    gauge = np.ascontiguousarray(f["gauge"])
    lat.set_gauge(gauge)
    lat.sw_term(gauge, s["kappa"], s["c_sw"])
    lat.sw_invert(0, s["mu"])
    sw, swi = lat.get_clover()
    assert rel_err(sw, c["sw"]) < TOL and rel_err(swi, c["sw_inv"]) < TOL
    # and the operator built on the device-computed blocks reproduces the reference's Qsw_pm_psi
    dk, dl = lat.field(np.ascontiguousarray(f["in"])), lat.field()
    lat.op("Qsw_pm_psi", dl, dk)
    assert rel_err(dl.download(), c["Qsw_pm_psi"]) < TOL
    lat.close()


@pytest.mark.parametrize("dims,mu", [((8, 6, 4, 12), 0.02), ((4, 4, 6, 2), 0.0), ((2, 2, 2, 2), 0.3)])
def test_device_sw_term_and_sw_invert_against_oracle(dims, mu):
    """Ragged extents (incl. the 2-site wrap where +mu and -mu neighbours coincide) and mu = 0, where only the first
    half of sw_inv is produced (clover_invert.c:225)."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = dims
    kappa, c_sw = 0.137, 1.57
    orc = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, threads=8)
    lat = Lattice(T, LX, LY, LZ, kappa=kappa, mu=mu)
    g = random_gauge(81, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    sw_ref = orc.sw_term(kappa, c_sw)
    swi_ref, fails = orc.sw_invert(sw_ref, 0, mu)
    assert fails == 0
    lat.sw_term(g, kappa, c_sw)
    for ieo in (1, 0):                       # the odd-site inverse is legal too; leave the even one in place
        lat.sw_invert(ieo, mu)
        sw, swi = lat.get_clover()
        ref_i, _ = orc.sw_invert(sw_ref, ieo, mu)
        n = orc.V if mu != 0.0 else orc.V // 2
        assert rel_err(sw, sw_ref) < TOL
        assert rel_err(swi[:n], ref_i[:n]) < TOL
        if mu == 0.0:
            assert not swi[n:].any()         # untouched, like the reference's second half
    orc.set_clover(sw_ref, swi_ref)
    k = random_spinor(82, orc.Vh)
    ref = orc.new_field(); orc.op("Qsw_pm_psi", ref, k.copy())
    dk, dl = lat.field(k), lat.field()
    lat.op("Qsw_pm_psi", dl, dk)
    assert rel_err(dl.download(), ref[:orc.Vh]) < TOL
    lat.close()


def test_device_sw_term_on_t_split_ranks():
    """Two T-slabs (nproc_t = 2) compute their clover blocks from their own gauge field + halo slabs and reproduce the
    slices of the unsplit lattice; set_clover accepts per-rank host arrays as well."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    from tmlqcd_amd import synthetic as syn
    T, L, world = 2, 4, 2
    Tg = T * world
    kappa, mu, c_sw = 0.13, 0.05, 1.3
    g = Oracle(Tg, L, L, L, kappa=kappa, mu=mu, threads=4)
    g.set_gauge(syn.gauge_field(5, Tg, L, L, L))
    sw_ref = g.sw_term(kappa, c_sw)
    swi_ref, _ = g.sw_invert(sw_ref, 0, mu)
    Vg = g.V
    for r in range(world):
        lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, nproc_t=world, proc_t=r)
        gr = syn.gauge_field(5, T, L, L, L, world, r)
        lat.set_gauge(gr)
        lat.sw_term(gr, kappa, c_sw)
        lat.sw_invert(0, mu)
        sw, swi = lat.get_clover()
        V, Vh = lat.V, lat.Vh
        assert rel_err(sw, sw_ref[r * V:(r + 1) * V]) < TOL
        assert rel_err(swi[:Vh], swi_ref[r * Vh:(r + 1) * Vh]) < TOL
        assert rel_err(swi[Vh:], swi_ref[Vg // 2 + r * Vh:Vg // 2 + (r + 1) * Vh]) < TOL
        lat.set_clover(np.ascontiguousarray(sw_ref[r * V:(r + 1) * V]), np.ascontiguousarray(swi))   # upload path on a split rank
        sw2, swi2 = lat.get_clover()
        assert np.array_equal(sw2, sw_ref[r * V:(r + 1) * V]) and np.array_equal(swi2, swi)
        lat.close()


@pytest.mark.parametrize("fused", [2, 0])
def test_clover_cg_with_reductions_fused_into_the_stencils(fused):
    """8^4 (V/2 % 256 == 0): cg_her on Qsw_pm_psi with pro = |Q_- p|^2 out of the second stencil's clover_gamma5 epilogue and
    r -= alpha A p, |r|^2 out of the fourth (cg_fused_dot = 2, the default) == the plain path == the oracle; the fp32 inner
    loops of mixed_cg_her / rg_mixed_cg_her take the same route."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T = L = 8
    kappa, mu = 0.13, 0.02
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, threads=8)
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
    g = random_gauge(75, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    sw = orc.sw_term(kappa, 1.3)
    swi, _ = orc.sw_invert(sw, 0, mu)
    orc.set_clover(sw, swi); lat.set_clover(sw, swi)
    lat.set_option("cg_fused_dot", fused)
    N = orc.Vh
    q = random_spinor(76, N)
    P = orc.new_field(); it_ref, hist_ref = orc.cg_her(P, q.copy(), 2000, 1e-20, 1, N, "Qsw_pm_psi")
    dq, dp = lat.field(q), lat.field()
    it, hist = lat.cg_her(dp, dq, 2000, 1e-20, 1, N, op="Qsw_pm_psi")
    assert abs(it - it_ref) <= 1 and rel_err(dp.download(), P[:N]) < 1e-9
    m = min(len(hist), len(hist_ref)) - 1
    assert np.allclose(hist[:m], hist_ref[:m], rtol=1e-6)
    for solver in (lambda: lat.mixed_cg_her(dp, dq, 5000, 1e-20, 1, N, op="Qsw_pm_psi")[0],
                   lambda: lat.rg_mixed_cg_her(dp, dq, 5000, 1e-20, 1, N, delta=0.1, op="Qsw_pm_psi")[0]):
        assert solver() > 0
        assert rel_err(dp.download(), P[:N]) < 1e-8
    lat.set_loopback(1)                      # split-phase path: reductions spread over boundary + interior kernels
    dp.zero()
    it2, _ = lat.cg_her(dp, dq, 2000, 1e-20, 1, N, op="Qsw_pm_psi")
    assert abs(it2 - it_ref) <= 1 and rel_err(dp.download(), P[:N]) < 1e-9
    assert lat.mixed_cg_her(dp, dq, 5000, 1e-20, 1, N, op="Qsw_pm_psi")[0] > 0 and rel_err(dp.download(), P[:N]) < 1e-8
    lat.set_loopback(0)
    lat.close()
