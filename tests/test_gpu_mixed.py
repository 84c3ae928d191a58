"""GPU: fp32 twins and the mixed-precision CG.  The fp32 reference path only exists in the
half-spinor build with a different operation order, so these rows are checked by properties: the fp32 operators
against the fp64 oracle at fp32 tolerance, mixed_cg_her by the true fp64 residual and against the fp64 solution."""
import numpy as np
import pytest

from tests.util import random_gauge, random_spinor

pytestmark = pytest.mark.gpu
TOL32 = 2e-6


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def setup():
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T = L = 8
    kappa, mu, theta = 0.13, 0.015, (1.0, 0.0, 0.0, 0.5)
    orc = Oracle(T, L, L, L, kappa=kappa, mu=mu, theta=theta, threads=8)
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu, theta=theta)
    g = random_gauge(61, orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    yield orc, lat
    lat.close()


@pytest.mark.parametrize("lds,recon", [(0, 18), (1, 18), (0, 12), (1, 12)])
def test_fp32_stencil_and_operator(setup, lds, recon):
    """lds=1: the per-wave LDS-staged instance (block 256); recon=12: the opt-in 12-real gauge read (third row rebuilt in fp32
    registers).  The fp32 links are read in their packed layout (float4 pairs of elements) in every case."""
    orc, lat = setup
    lat.set_option("lds32", lds)
    lat.set_option("block", 256 if lds else 0)
    lat.set_option("gauge_recon", recon)
    N = orc.Vh
    k32 = random_spinor(1, N).astype(np.float32)
    k = k32.astype(np.float64)
    ref = orc.new_field()
    dk, dl = lat.field32(k32), lat.field32()
    for ieo in (0, 1):
        orc.Hopping_Matrix(ieo, ref, k)
        lat.Hopping_Matrix_32(ieo, dl, dk)
        assert rel(dl.download().astype(np.float64), ref[:N]) < TOL32
    orc.op("Qtm_pm_psi", ref, k.copy())
    lat.Qtm_pm_psi_32(dl, dk)
    assert rel(dl.download().astype(np.float64), ref[:N]) < 4 * TOL32
    lat.set_loopback(1)                                   # split-phase path in fp32 (faces packed as float2)
    for ieo in (0, 1):
        orc.Hopping_Matrix(ieo, ref, k)
        lat.Hopping_Matrix_32(ieo, dl, dk)
        assert rel(dl.download().astype(np.float64), ref[:N]) < TOL32
    lat.set_loopback(0)
    if recon == 12:                                       # and the mixed solver on top of it
        q = random_spinor(2, N)
        dq, dp = lat.field(q), lat.field()
        it, outer = lat.mixed_cg_her(dp, dq, 5000, 1e-20, 1, N)
        full = orc.new_field(); full[:N] = dp.download()
        chk = orc.new_field(); orc.op("Qtm_pm_psi", chk, full)
        assert it > 0 and ((chk[:N] - q) ** 2).sum() / (q ** 2).sum() <= 1e-20
        dq.free(); dp.free()
    lat.set_option("gauge_recon", 18)
    lat.set_option("lds32", 0); lat.set_option("block", 0)
    dk.free(); dl.free()


@pytest.mark.parametrize("dims", [(4, 4, 4, 6), (2, 2, 2, 4), (4, 6, 2, 12), (2, 2, 2, 2)])
def test_fp32_stencil_ragged(dims):
    """LZ/2 odd (6, 2) falls back to one site per thread; LZ/2 even (4, 12) uses site pairs with every row wrap."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    T, LX, LY, LZ = dims
    orc = Oracle(T, LX, LY, LZ, kappa=0.14, mu=0.03, theta=(1.0, 0.3, -0.2, 0.5))
    lat = Lattice(T, LX, LY, LZ, kappa=0.14, mu=0.03, theta=(1.0, 0.3, -0.2, 0.5))
    g = random_gauge(sum(dims), orc.VPR)
    orc.set_gauge(g); lat.set_gauge(g)
    N = orc.Vh
    k32 = random_spinor(3, N).astype(np.float32)
    ref = orc.new_field()
    dk, dl = lat.field32(k32), lat.field32()
    for ieo in (0, 1):
        orc.Hopping_Matrix(ieo, ref, k32.astype(np.float64))
        lat.Hopping_Matrix_32(ieo, dl, dk)
        assert rel(dl.download().astype(np.float64), ref[:N]) < TOL32
    orc.op("Qtm_pm_psi", ref, k32.astype(np.float64))
    lat.Qtm_pm_psi_32(dl, dk)
    assert rel(dl.download().astype(np.float64), ref[:N]) < 4 * TOL32
    lat.close()


def test_fp32_linalg_and_conversions(setup):
    orc, lat = setup
    N = orc.Vh
    a64, b64 = random_spinor(2, N), random_spinor(3, N)
    da, d32, db32, back = lat.field(a64), lat.field32(), lat.field32(b64.astype(np.float32)), lat.field()
    lat.assign_to_32(d32, da, N)
    assert np.array_equal(d32.download(), a64.astype(np.float32))
    lat.assign_to_64(back, d32, N)
    assert np.array_equal(back.download(), a64.astype(np.float32).astype(np.float64))
    a32, b32 = a64.astype(np.float32).astype(np.float64), b64.astype(np.float32).astype(np.float64)
    assert abs(lat.square_norm_32(d32, N) - (a32 ** 2).sum()) <= 1e-12 * (a32 ** 2).sum()     # double accumulation
    assert abs(lat.scalar_prod_r_32(d32, db32, N) - (a32 * b32).sum()) <= 1e-10 * (a32 ** 2).sum()
    lat.assign_add_mul_r_32(d32, db32, 0.5, N)
    assert rel(d32.download().astype(np.float64), a32 + 0.5 * b32) < TOL32
    lat.assign_mul_add_r_32(d32, -0.25, db32, N)
    assert rel(d32.download().astype(np.float64), -0.25 * (a32 + 0.5 * b32) + b32) < TOL32
    for f in (da, d32, db32, back):
        f.free()


@pytest.mark.parametrize("fused", [2, 1, 0])
def test_mixed_cg_her_reaches_fp64_residual(setup, fused):
    orc, lat = setup
    lat.set_option("cg_fused_dot", fused)
    N = orc.Vh
    q = random_spinor(4, N)
    eps_sq = 1e-20
    dq, dp = lat.field(q), lat.field()
    it, outer = lat.mixed_cg_her(dp, dq, 5000, eps_sq, 1, N)
    assert it > 0 and outer >= 2                  # fp32 alone cannot reach 1e-10: at least one fp64 restart
    sol = dp.download()
    full = orc.new_field(); full[:N] = sol
    chk = orc.new_field(); orc.op("Qtm_pm_psi", chk, full)
    res = ((chk[:N] - q) ** 2).sum() / (q ** 2).sum()
    assert res <= eps_sq                          # true residual, fp64 on the CPU (operator.c:379-384)
    P = orc.new_field(); it64, _ = orc.cg_her(P, q.copy(), 5000, eps_sq, 1, N)
    assert rel(sol, P[:N]) < 1e-8
    assert it < 2.0 * it64 + 20                   # same order of work as the fp64 CG
    # same right-hand side through the fp64 solver on the GPU for reference
    dp2 = lat.field(); it_gpu, _ = lat.cg_her(dp2, dq, 5000, eps_sq, 1, N)
    assert rel(sol, dp2.download()) < 1e-8
    lat.set_option("cg_fused_dot", 2)
    for f in (dq, dp, dp2):
        f.free()


def test_mixed_cg_loose_tolerance_single_outer(setup):
    orc, lat = setup
    N = orc.Vh
    dq, dp = lat.field(random_spinor(5, N)), lat.field()
    it, outer = lat.mixed_cg_her(dp, dq, 5000, 1e-6, 1, N, innereps=1e-7)
    assert it > 0 and outer == 1
    dq.free(); dp.free()


def test_fp32_operators_against_reference_fp32_fixture():
    """tests/golden/ref_hs_fields_4x4.npz: Hopping_Matrix_32 / Qtm_pm_psi_32 / linalg_32 outputs of the reference's
    half-spinor build (the only configuration that has the fp32 twins).  Different operation order => fp32 tolerance."""
    import json
    import os
    from tmlqcd_amd import Lattice
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    f = np.load(os.path.join(gold, "ref_fields_4x4.npz"))
    h = np.load(os.path.join(gold, "ref_hs_fields_4x4.npz"))
    s = json.load(open(os.path.join(gold, "ref_hs_scalars_4x4.json")))
    lat = Lattice(4, 4, 4, 4, kappa=s["kappa"], mu=s["mu"])
    lat.set_gauge(np.ascontiguousarray(f["gauge"]))
    N = lat.Vh
    d64, d32, a, b = lat.field(np.ascontiguousarray(f["in"])), lat.field32(), lat.field32(), lat.field32()
    lat.assign_to_32(d32, d64, N)
    assert np.array_equal(d32.download(), h["in32"])
    lat.Hopping_Matrix_32(0, a, d32)
    assert rel(a.download().astype(np.float64), h["Heo32"].astype(np.float64)) < TOL32
    lat.Hopping_Matrix_32(1, b, a)
    assert rel(b.download().astype(np.float64), h["HoeHeo32"].astype(np.float64)) < TOL32
    lat.Qtm_pm_psi_32(b, d32)
    assert rel(b.download().astype(np.float64), h["Qtm_pm_psi_32"].astype(np.float64)) < 2 * TOL32
    assert abs(lat.square_norm_32(d32, N) - s["square_norm_32_in"]) <= 1e-6 * s["square_norm_32_in"]
    assert abs(lat.scalar_prod_r_32(d32, b, N) - s["scalar_prod_r_32_in_Qpm"]) <= 1e-5 * abs(s["scalar_prod_r_32_in_Qpm"])
    lat.close()


# ------------------------------------------------------------------ rg_mixed_cg_her (solver/rg_mixed_cg_her.c:180)
def _rg_fixture():
    import json, os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    f = np.load(os.path.join(gold, "ref_fields_4x4.npz"))
    g = np.load(os.path.join(gold, "ref_rg_fields_4x4.npz"))
    s = json.load(open(os.path.join(gold, "ref_rg_scalars_4x4.json")))
    return f, g, s


def test_rg_mixed_cg_her_against_reference_run():
    """Same gauge field, source, eps and mcg_delta as the reference's own rg_mixed_cg_her (default half-spinor
    build, oracle/make_golden.py rgfull).  fp32 rounding differs (full spinors vs half spinors, reduction order),
    so restart points may move by an iteration or two; the solution must agree to fp64-solver accuracy."""
    from oracle.oraclebind import Oracle
    from tmlqcd_amd import Lattice
    f, g, s = _rg_fixture()
    lat = Lattice(4, 4, 4, 4, kappa=s["kappa"], mu=s["mu"])
    orc = Oracle(4, 4, 4, 4, kappa=s["kappa"], mu=s["mu"])
    gauge = np.ascontiguousarray(f["gauge"])
    lat.set_gauge(gauge); orc.set_gauge(gauge)
    N = lat.Vh
    src = np.ascontiguousarray(f["in"])
    dq, dp = lat.field(src), lat.field()
    for run in s["runs"]:
        dp.upload(np.full_like(src, 3.0))               # "initial guess currently not supported": P is zeroed inside
        it, (n_out, n_sp, n_dp) = lat.rg_mixed_cg_her(dp, dq, 2000, s["eps_sq"], s["rel_prec"], N, delta=run["delta"])
        if run["iters"] < 0:                            # delta = 0.5: N_outer exhausted -> convergence failure, like the reference
            assert it == -1 and n_dp > 0, (run, it, n_out, n_sp, n_dp)
            continue
        assert it > 0 and it == n_out + n_sp + n_dp
        assert abs(it - run["iters"]) <= max(3, run["iters"] // 8), (run, it, n_out, n_sp, n_dp)
        sol = dp.download()
        assert rel(sol, g["solution_delta_%g" % run["delta"]]) < 1e-8
        full = orc.new_field(); full[:N] = sol
        chk = orc.new_field(); orc.op("Qtm_pm_psi", chk, full)
        assert ((chk[:N] - src) ** 2).sum() / (src ** 2).sum() <= s["eps_sq"]
    lat.close()


def test_rg_mixed_cg_her_8x8(setup):
    orc, lat = setup
    N = orc.Vh
    q = random_spinor(14, N)
    dq, dp = lat.field(q), lat.field()
    P = orc.new_field(); it64, _ = orc.cg_her(P, q.copy(), 5000, 1e-20, 1, N)
    for delta in (5.0e-5, 0.1):
        it, (n_out, n_sp, n_dp) = lat.rg_mixed_cg_her(dp, dq, 5000, 1e-20, 1, N, delta=delta)
        assert it > 0 and n_out >= 2 and it == n_out + n_sp + n_dp
        assert rel(dp.download(), P[:N]) < 1e-8
        assert n_sp + n_dp < 2.0 * it64 + 20
    # iteration cap: max_iter smaller than needed -> -1 like rg_mixed_cg_her.c:303-304
    it, _ = lat.rg_mixed_cg_her(dp, dq, 10, 1e-20, 1, N, delta=0.1)
    assert it == -1
    # absolute precision branch (rel_prec = 0)
    it, _ = lat.rg_mixed_cg_her(dp, dq, 5000, 1e-14, 0, N, delta=0.1)
    full = orc.new_field(); full[:N] = dp.download()
    chk = orc.new_field(); orc.op("Qtm_pm_psi", chk, full)
    assert it > 0 and ((chk[:N] - q) ** 2).sum() <= 1e-14
    dq.free(); dp.free()
