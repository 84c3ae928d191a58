"""CPU: the oracle against the reference's own known answers.

1. tests/test_linalg_spinor.c of the reference (literal inputs -> literal results, EPS=1e-15).
2. SURVEY.md §8c / tests/golden/ref_scalars_*.json: norms, a site value and cg_her iteration counts
   produced by the reference object code on RANLUX-seeded inputs.
3. tests/golden/ref_fields_4x4.npz: full per-site outputs of the reference at 4^4.
"""
import json
import os

import numpy as np
import pytest

from oracle.oraclebind import Oracle
from tests.util import TOL, rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sp(rows):
    a = np.zeros((len(rows), 4, 3, 2))
    for i, r in enumerate(rows):
        a[i] = np.array(r, dtype=np.float64).reshape(4, 3, 2)
    return a


@pytest.fixture(scope="module")
def ka():
    return json.load(open(os.path.join(GOLD, "linalg_known_answers.json")))


@pytest.fixture(scope="module")
def orc():
    return Oracle(4, 4, 4, 4)


def test_linalg_known_answers(orc, ka):
    EPS = 1e-12  # the reference compares printed 17-digit literals with EPS=1e-15 one-sided; we are two-sided
    R, S = _sp(ka["R"]), _sp(ka["S"])
    assert abs(orc.scalar_prod_r(R, S, 2) - ka["scalar_prod_r_R_S"]) < EPS
    assert abs(orc.square_norm(R, 2) - ka["square_norm_R"]) < EPS
    Q = np.zeros_like(R)
    orc.diff(Q, R, S, 2)
    assert abs(orc.square_norm(Q, 2) - ka["diff_R_minus_S_norm"]) < EPS
    assert abs(Q[0, 0, 0, 0] - ka["diff_Q0_s0c0_re"]) < EPS and abs(Q[1, 2, 1, 1] - ka["diff_Q1_s2c1_im"]) < EPS
    A = R.copy()
    orc.assign_add_mul_r(A, S, ka["c"], 2)
    assert abs(orc.square_norm(A, 2) - ka["assign_add_mul_r_norm"]) < EPS
    assert abs(A[0, 0, 0, 0] - ka["assign_add_mul_r_R0_s0c0_re"]) < EPS
    assert abs(A[1, 2, 1, 1] - ka["assign_add_mul_r_R1_s2c1_im"]) < EPS
    B = R.copy()
    orc.assign_mul_add_r(B, ka["c"], S, 2)
    assert abs(orc.square_norm(B, 2) - ka["assign_mul_add_r_norm"]) < EPS
    assert abs(B[0, 0, 0, 0] - ka["assign_mul_add_r_R0_s0c0_re"]) < EPS
    assert abs(B[1, 2, 1, 1] - ka["assign_mul_add_r_R1_s2c1_im"]) < EPS
    C = R.copy()
    n = orc.assign_mul_add_r_and_square(C, ka["c"], S, 2)
    assert abs(n - ka["assign_mul_add_r_norm"]) < EPS and np.array_equal(C, B)


@pytest.fixture(scope="module")
def gold4():
    f = np.load(os.path.join(GOLD, "ref_fields_4x4.npz"))
    s = json.load(open(os.path.join(GOLD, "ref_scalars_4x4.json")))
    o = Oracle(4, 4, 4, 4, kappa=s["kappa"], mu=s["mu"])
    o.set_gauge(f["gauge"])
    return o, f, s


def test_golden_4x4_hopping_bit_exact(gold4):
    o, f, s = gold4
    N = o.Vh
    assert np.array_equal(o.eo2lexic(), f["eo2lexic"])
    l1, l2 = o.new_field(), o.new_field()
    o.Hopping_Matrix(0, l1, f["in"])
    o.Hopping_Matrix(1, l2, l1)
    # the restatement follows the reference's operation order: bit-for-bit equal
    assert np.array_equal(l1[:N], f["Heo"]) and np.array_equal(l2[:N], f["HoeHeo"])
    assert o.square_norm(f["in"], N) == s["norm_in"]
    assert o.square_norm(l1, N) == s["norm_Heo"] and o.square_norm(l2, N) == s["norm_HoeHeo"]
    assert list(l2[0, 0, 0]) == s["HoeHeo_site0_s0c0"]
    c = complex(*s["cfactor"])
    o.tm_times_Hopping_Matrix(1, l2, l1, c)
    assert np.array_equal(l2[:N], f["tm_times_OE_of_Heo"])
    o.tm_sub_Hopping_Matrix(1, l2, f["in"], l1, c)
    assert np.array_equal(l2[:N], f["tm_sub_OE_p_in_k_Heo"])


@pytest.mark.parametrize("name", ["Qtm_pm_psi", "Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi"])
def test_golden_4x4_operators(gold4, name):
    o, f, s = gold4
    N = o.Vh
    out = o.new_field()
    o.op(name, out, np.ascontiguousarray(f["in"]))
    assert np.array_equal(out[:N], f[name])


SYM_OPS = ["Qtm_plus_sym_psi", "Qtm_minus_sym_psi", "Mtm_plus_sym_psi", "Mtm_minus_sym_psi", "Mtm_plus_sym_dagg_psi",
           "Qtm_pm_sym_psi"]


@pytest.mark.parametrize("name", SYM_OPS)
def test_golden_4x4_symmetric_preconditioning_family(gold4, name):
    """tm_operators.c:186-364 (operators of the non-hermitian solvers, invert_eo.c:177-280)."""
    o, f, s = gold4
    g = np.load(os.path.join(GOLD, "ref_sym_fields_4x4.npz"))
    out = o.new_field()
    o.op(name, out, np.ascontiguousarray(f["in"]))
    assert np.array_equal(out[:o.Vh], g[name])


def test_golden_4x4_fermion_force(gold4):
    """deriv_Sb.c:401: two calls with swapped roles accumulated, as oracle/make_golden.py force made them."""
    o, f, s = gold4
    want = np.load(os.path.join(GOLD, "ref_force_4x4.npz"))["derivative"]
    N = o.Vh
    a, b = o.new_field(), o.new_field()
    a[:N] = f["in"]; b[:N] = f["Heo"]
    df = np.zeros((o.VPR, 4, 8))
    o.deriv_Sb(1, a, b, df, 0.5)
    o.deriv_Sb(0, b, a, df, -0.25)
    assert np.array_equal(df[:o.V], want)


def test_golden_4x4_M_full_and_D_psi(gold4):
    o, f, s = gold4
    N = o.Vh
    en, on = o.new_field(), o.new_field()
    o.M_full(en, on, f["in"], f["Heo"])
    assert np.array_equal(en[:N], f["M_full_even"]) and np.array_equal(on[:N], f["M_full_odd"])
    P = np.zeros_like(f["D_psi_in_lexic"])
    o.D_psi(P, f["D_psi_in_lexic"])
    assert rel_err(P, f["D_psi_out_lexic"]) < 1e-15
    # D_psi is M_full in lexicographic order (D = (1 + i mu g5) - H): cross-check of the two code paths
    e2l = f["eo2lexic"]
    assert rel_err(P[e2l[:N]], f["M_full_even"]) < TOL and rel_err(P[e2l[N:2 * N]], f["M_full_odd"]) < TOL


def test_golden_4x4_cg(gold4):
    o, f, s = gold4
    N = o.Vh
    P = o.new_field()
    it, hist = o.cg_her(P, np.ascontiguousarray(f["in"]), 1000, 1e-20, 1, N)
    assert it == s["cg_iters"] == 25
    assert np.array_equal(P[:N], f["cg_solution"])
    assert abs(o.square_norm(P, N) - s["cg_sol_norm"]) <= 1e-13 * s["cg_sol_norm"]


def test_survey_known_scalars_are_in_the_fixtures():
    """The numbers quoted in SURVEY.md §8c, recorded by the survey session from the reference."""
    s4 = json.load(open(os.path.join(GOLD, "ref_scalars_4x4.json")))
    s8 = json.load(open(os.path.join(GOLD, "ref_scalars_8x8.json")))
    def eq(a, b):  # SURVEY prints 16 significant digits
        return abs(a - b) <= 2e-15 * abs(b)
    assert eq(s4["norm_in"], 1.557156583000509e+03) and eq(s4["norm_Heo"], 3.771468883550644e+02)
    assert eq(s4["norm_HoeHeo"], 9.310810785930991e+01) and s4["cg_iters"] == 25
    assert eq(s4["HoeHeo_site0_s0c0"][0], 3.822793515307792e-02) and eq(s4["HoeHeo_site0_s0c0"][1], 2.194152107038742e-01)
    assert eq(s8["norm_in"], 2.487157888943338e+04) and eq(s8["norm_Heo"], 6.168649177022544e+03)
    assert eq(s8["norm_HoeHeo"], 1.602210010255744e+03) and s8["cg_iters"] == 36
    assert eq(s8["HoeHeo_site0_s0c0"][0], 2.112889071498668e-02) and eq(s8["HoeHeo_site0_s0c0"][1], 1.575582919029603e-01)


def test_default_halfspinor_build_of_the_reference_agrees(gold4):
    """tests/golden/ref_hs_fields_4x4.npz was produced by the reference's DEFAULT configuration
    (_USE_HALFSPINOR, operator/halfspinor_body.c): an independently written implementation of the same
    operator.  The oracle (generic body) must reproduce it; its fp32 twins agree at fp32 accuracy."""
    o, f, s = gold4
    h = np.load(os.path.join(GOLD, "ref_hs_fields_4x4.npz"))
    N = o.Vh
    l1, l2, q = o.new_field(), o.new_field(), o.new_field()
    o.Hopping_Matrix(0, l1, f["in"]); o.Hopping_Matrix(1, l2, l1)
    o.op("Qtm_pm_psi", q, np.ascontiguousarray(f["in"]))
    assert rel_err(l1[:N], h["hs_Heo"]) < 1e-15 and rel_err(l2[:N], h["hs_HoeHeo"]) < 1e-15
    assert rel_err(q[:N], h["hs_Qtm_pm_psi"]) < 1e-15
    assert np.array_equal(h["in32"], f["in"].astype(np.float32))                    # assign_to_32 = plain rounding
    assert rel_err(h["Heo32"].astype(np.float64), l1[:N]) < 1e-6
    assert rel_err(h["Qtm_pm_psi_32"].astype(np.float64), q[:N]) < 1e-6


def test_golden_4x4_clover(gold4):
    """Clover twisted mass: sw / sw_inv from the reference's sw_term / sw_invert and the
    reference's clover_inv, clover_gamma5, Qsw_pm_psi, Msw_plus_psi and cg_her(Qsw_pm_psi) on them."""
    o, f, s = gold4
    c = np.load(os.path.join(GOLD, "ref_clover_fields_4x4.npz"))
    cs = json.load(open(os.path.join(GOLD, "ref_clover_scalars_4x4.json")))
    N = o.Vh
    mu = cs["mu"]
    o.set_clover(c["sw"], c["sw_inv"])
    # sw is hermitian block-wise and sw_inv really is the inverse of (1 + T +- i mu g5) on the even sites
    sw, swi = c["sw"][..., 0] + 1j * c["sw"][..., 1], c["sw_inv"][..., 0] + 1j * c["sw_inv"][..., 1]
    x = f["eo2lexic"][5]                     # an even site, e/o index 5
    for chi, sgn in ((0, +1), (1, -1)):
        A, B, Cc = sw[x, 0, chi], sw[x, 1, chi], sw[x, 2, chi]
        M = np.block([[A, B], [B.conj().T, Cc]]) + 1j * sgn * mu * np.eye(6)
        Minv = np.block([[swi[5, 0, chi], swi[5, 1, chi]], [swi[5, 3, chi], swi[5, 2, chi]]])
        assert np.abs(M @ Minv - np.eye(6)).max() < 1e-13
    # the oracle's own sw_term / sw_invert reproduce both arrays from the fixture's gauge field, bit for bit
    sw_o = o.sw_term(cs["kappa"], cs["c_sw"])
    assert np.array_equal(sw_o, c["sw"])
    swi_o, fails = o.sw_invert(sw_o, 0, mu)
    assert fails == 0 and np.array_equal(swi_o, c["sw_inv"])
    k = np.ascontiguousarray(f["in"])
    for sign, key in ((-1, "clover_inv_minus"), (+1, "clover_inv_plus")):
        l = o.new_field(); l[:N] = k
        o.clover_inv(l, sign, mu)
        assert np.array_equal(l[:N], c[key])
    h = o.new_field(); o.Hopping_Matrix(1, h, k)
    l = o.new_field(); o.clover_gamma5(1, l, k, h, -mu)
    assert np.array_equal(l[:N], c["clover_gamma5_OO_in_Hoe"])
    q = o.new_field(); o.op("Qsw_pm_psi", q, k.copy())
    assert np.array_equal(q[:N], c["Qsw_pm_psi"])
    o.op("Msw_plus_psi", q, k.copy())
    assert np.array_equal(q[:N], c["Msw_plus_psi"])
    P = o.new_field()
    it, _ = o.cg_her(P, k.copy(), 1000, 1e-20, 1, N, "Qsw_pm_psi")
    assert it == cs["cg_iters"] and np.array_equal(P[:N], c["cg_solution"])
