"""CPU: oracle/tm_oracle.c (our restatement) against oracle/_ref/libtmref.so (the reference's own
sources compiled in place) on identical RANLUX-seeded inputs.  One subprocess per lattice because
the reference keeps its state in C globals.  Skipped when oracle/_ref has not been built (it is
built wherever /root/reference exists and travels to the GPU box as a prebuilt .so)."""
import os
import subprocess
import sys

import pytest

from oracle import refbind

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle.refbind import RefLattice
from oracle.oraclebind import Oracle
T, LX, LY, LZ = %(dims)s
kappa, mu, theta = 0.131, 0.017, (1.0, 0.5, -0.25, 0.125)
r = RefLattice(T, LX, LY, LZ, kappa=kappa, mu=mu, nfields=16)  # fields 13..15 = DUM_MATRIX scratch (tm_operators.c:173-176)
r.set_theta(*theta)
r.random_fields(4711)
o = Oracle(T, LX, LY, LZ, kappa=kappa, mu=mu, theta=theta)
N, V = r.V // 2, r.V
lib, sp = r.lib, r.sp
assert np.array_equal(o.eo2lexic(), r.eo2lexic()), "eo2lexic"
assert np.array_equal(o.hi()[:V], r.hi()), "g_hi"
o.set_gauge(r.gauge().copy())
k = r.spinor(0, N).copy()
def same(a, i, what):
    assert np.array_equal(a[:N], r.spinor(i, N)), what
l, l2, q = o.new_field(), o.new_field(), o.new_field()
for ieo in (0, 1):
    o.Hopping_Matrix(ieo, l, k); lib.Hopping_Matrix(ieo, sp(1), sp(0)); same(l, 1, "Hopping_Matrix %%d" %% ieo)
    c = -0.37 + 0.91j
    o.tm_times_Hopping_Matrix(ieo, l2, k, c); lib.tm_times_Hopping_Matrix(ieo, sp(2), sp(0), c.real, c.imag); same(l2, 2, "tm_times")
    o.tm_sub_Hopping_Matrix(ieo, l2, l, k, c); lib.tm_sub_Hopping_Matrix(ieo, sp(2), sp(1), sp(0), c.real, c.imag); same(l2, 2, "tm_sub")
for name in ("Qtm_pm_psi", "Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi", "Qtm_plus_sym_psi",
             "Qtm_minus_sym_psi", "Mtm_plus_sym_psi", "Mtm_minus_sym_psi", "Mtm_plus_sym_dagg_psi", "Qtm_pm_sym_psi"):
    o.op(name, q, k); getattr(lib, name)(sp(3), sp(0)); same(q, 3, name)
# in-place Qtm_minus_psi as invert_eo.c:270 calls it
a = k.copy(); b = o.new_field(); b[:N] = a
o.op("Qtm_minus_psi", b, b); lib.assign(sp(4), sp(0), N); lib.Qtm_minus_psi(sp(4), sp(4)); same(b, 4, "Qtm_minus_psi in place")
o.H_eo_tm_inv_psi(q, k, 1, -1.); lib.H_eo_tm_inv_psi(sp(3), sp(0), 1, -1.); same(q, 3, "H_eo_tm_inv_psi")
# site-diagonal ops
for sign in (+1., -1.):
    o.assign_mul_one_pm_imu_inv(q, k, sign, N); lib.assign_mul_one_pm_imu_inv(sp(3), sp(0), sign, N); same(q, 3, "assign_mul_one_pm_imu_inv")
    o.assign_mul_one_pm_imu(q, k, sign, N); lib.assign_mul_one_pm_imu(sp(3), sp(0), sign, N); same(q, 3, "assign_mul_one_pm_imu")
    o.mul_one_pm_imu_inv(q, sign, N); lib.mul_one_pm_imu_inv(sp(3), sign, N); same(q, 3, "mul_one_pm_imu_inv")
    o.mul_one_pm_imu_sub_mul(l2, k, q, sign, N); lib.mul_one_pm_imu_sub_mul(sp(2), sp(0), sp(3), sign, N); same(l2, 2, "mul_one_pm_imu_sub_mul")
    o.mul_one_pm_imu_sub_mul_gamma5(l2, k, q, sign); lib.mul_one_pm_imu_sub_mul_gamma5(sp(2), sp(0), sp(3), sign); same(l2, 2, "..._gamma5")
o.gamma5(l2, k, N); lib.gamma5(sp(2), sp(0), N); same(l2, 2, "gamma5")
# linalg (1 thread => identical Kahan order)
assert o.square_norm(k, N) == lib.square_norm(sp(0), N, 0)
assert o.scalar_prod_r(k, q, N) == lib.scalar_prod_r(sp(0), sp(3), N, 0)
x = k.copy(); o.assign_add_mul_r(x, q, 0.37, N); lib.assign(sp(5), sp(0), N); lib.assign_add_mul_r(sp(5), sp(3), 0.37, N); same(x, 5, "assign_add_mul_r")
o.assign_mul_add_r(x, -1.2, q, N); lib.assign_mul_add_r(sp(5), -1.2, sp(3), N); same(x, 5, "assign_mul_add_r")
n1 = o.assign_mul_add_r_and_square(x, 0.6, q, N); n2 = lib.assign_mul_add_r_and_square(sp(5), 0.6, sp(3), N, 0)
assert n1 == n2; same(x, 5, "assign_mul_add_r_and_square")
o.diff(x, k, q, N); lib.diff(sp(5), sp(0), sp(3), N); same(x, 5, "diff")
o.add(x, k, q, N); lib.add(sp(5), sp(0), sp(3), N); same(x, 5, "add")
o.mul_r(x, -0.73, q, N); lib.mul_r(sp(5), -0.73, sp(3), N); same(x, 5, "mul_r")
# M_full and D_psi
en, on = o.new_field(), o.new_field()
o.M_full(en, on, k, q); lib.M_full(sp(6), sp(7), sp(0), sp(3)); same(en, 6, "M_full even"); same(on, 7, "M_full odd")
lex = np.random.default_rng(5).standard_normal((V, 4, 3, 2)); r.spinor(8, V)[:] = lex
P = np.zeros_like(lex); o.D_psi(P, lex); lib.D_psi(sp(9), sp(8))
assert np.abs(P - r.spinor(9, V)).max() <= 1e-15 * np.abs(P).max(), "D_psi"
# cg_her
Pc = o.new_field(); it, hist = o.cg_her(Pc, k.copy(), 500, 1e-18, 1, N)
r.spinor(10)[:] = 0; lib.assign(sp(11), sp(0), N)
it2 = lib.cg_her(sp(10), sp(11), 500, 1e-18, 1, N, r.fnptr("Qtm_pm_psi"))
assert it == it2, (it, it2); same(Pc, 10, "cg solution")
# fermion force, hopping part (deriv_Sb.c:401): both parities accumulated into one derivative field
df = np.zeros((o.VPR, 4, 8))
lsp, ksp = o.new_field(), o.new_field(); lsp[:N] = r.spinor(1, N); ksp[:N] = r.spinor(2, N)
for ieo, fac in ((0, 0.7), (1, -1.3)):
    r.deriv_Sb(ieo, 1, 2, fac); o.deriv_Sb(ieo, lsp, ksp, df, fac)
assert np.array_equal(df[:V], r.derivative()) and np.abs(df).max() > 0, "deriv_Sb"
# clover term and its inverse (operator/clover_term.c:88, operator/clover_invert.c:170): the reference's sw_term /
# sw_invert(EE, mu) on its own gauge field vs the restatement, +mu and -mu sets
sw_ref, swi_ref = r.clover(1.37, 0.02)
sw = o.sw_term(kappa, 1.37)
assert np.array_equal(sw, sw_ref), "sw_term"
swi, fails = o.sw_invert(sw, 0, 0.02)
assert fails == 0 and np.array_equal(swi, swi_ref), "sw_invert"
# the e/o clover operator family on those blocks (operator/clovertm_operators.c:96-268, assign_mul_one_sw_pm_imu_inv_block_body.c)
import ctypes as C
o.set_clover(sw, swi); o.set_mu(0.02)          # tmref_clover left g_mu = 0.02 in the reference
for name in ("Qsw_pm_psi", "Qsw_psi", "Qsw_plus_psi", "Qsw_minus_psi", "Qsw_sq_psi", "Msw_psi", "Msw_plus_psi", "Msw_minus_psi"):
    fn = getattr(lib, name); fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = None
    o.op(name, q, k); fn(sp(3), sp(0)); same(q, 3, name)
C.c_double.in_dll(lib, "g_mu3").value = 0.07; o.set_mu3(0.07)     # odd-odd twist g_mu + g_mu3 (clovertm_operators.c:208,216,238,243,258,265)
for name in ("Qsw_pm_psi", "Qsw_psi", "Qsw_plus_psi", "Qsw_minus_psi", "Qsw_sq_psi", "Msw_psi", "Msw_plus_psi", "Msw_minus_psi"):
    o.op(name, q, k); getattr(lib, name)(sp(3), sp(0)); same(q, 3, name + " with g_mu3")
C.c_double.in_dll(lib, "g_mu3").value = 0.0; o.set_mu3(0.0)
b = o.new_field(); b[:N] = k
o.op("Qsw_minus_psi", b, b); lib.assign(sp(4), sp(0), N); lib.Qsw_minus_psi(sp(4), sp(4)); same(b, 4, "Qsw_minus_psi in place (invert_clover_eo.c:128)")
for name in ("assign_mul_one_sw_pm_imu", "assign_mul_one_sw_pm_imu_inv"):
    fn = getattr(lib, name); fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double]; fn.restype = None
for ieo in (0, 1):
    o.assign_mul_one_sw_pm_imu(ieo, q, k, 0.02); lib.assign_mul_one_sw_pm_imu(ieo, sp(3), sp(0), 0.02); same(q, 3, "assign_mul_one_sw_pm_imu")
o.assign_mul_one_sw_pm_imu_inv(0, q, k, 0.02); lib.assign_mul_one_sw_pm_imu_inv(0, sp(3), sp(0), 0.02); same(q, 3, "assign_mul_one_sw_pm_imu_inv")
# clover part of the force (operator/clover_deriv.c:72,252; operator/clover_accumulate_deriv.c:58)
lib.sw_spinor_eo.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double]; lib.sw_deriv.argtypes = [C.c_int, C.c_double]
lib.tmref_sw_all.argtypes = [C.c_double, C.c_double]
lib.tmref_swpm_zero()
lib.sw_spinor_eo(0, sp(0), sp(1), 0.7); lib.sw_spinor_eo(1, sp(1), sp(0), -0.3); lib.sw_deriv(0, 0.02)
swm_r, swp_r = r.swpm()
swm = np.zeros((V, 4, 3, 3, 2)); swp = np.zeros_like(swm)
o.sw_spinor_eo(0, swm, swp, k, l, 0.7); o.sw_spinor_eo(1, swm, swp, l, k, -0.3); o.sw_deriv(0, swm, swp, 0.02)
assert np.array_equal(swm, swm_r) and np.array_equal(swp, swp_r) and np.abs(swp).max() > 0, "sw_spinor_eo / sw_deriv"
d0 = r.derivative().copy()
lib.tmref_sw_all(kappa, 1.37)
dfc = np.zeros((o.VPR, 4, 8)); o.sw_all(dfc, swm, swp, kappa, 1.37)
assert np.array_equal(dfc[:V], r.derivative() - d0) or np.abs(dfc[:V] - (r.derivative() - d0)).max() < 1e-13 * np.abs(dfc).max(), "sw_all"
lib.Msw_full.argtypes = [C.c_void_p] * 4; lib.Msw_full.restype = None
en, on = o.new_field(), o.new_field()
o.Msw_full(en, on, k, l); lib.Msw_full(sp(6), sp(7), sp(0), sp(1)); same(en, 6, "Msw_full even"); same(on, 7, "Msw_full odd")
print("OK", T, LX, LY, LZ, "cg iters", it)
'''


@pytest.mark.skipif(not refbind.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("dims", [(4, 4, 4, 4), (6, 4, 4, 4), (4, 6, 4, 8), (8, 8, 8, 8)])
def test_oracle_matches_reference_bit_for_bit(dims):
    code = CHILD % {"root": ROOT, "dims": repr(dims)}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


MD_CHILD = r'''
import sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
from oracle.refbind import RefLattice
from oracle.oraclebind import Oracle
T, LX, LY, LZ = %(dims)s
r = RefLattice(T, LX, LY, LZ, kappa=0.125, mu=0.01, nfields=8, hs=True)
r.random_fields(991)
o = Oracle(T, LX, LY, LZ)
mom = np.random.default_rng(5).standard_normal((r.V, 4, 8)) * 3.0      # large momenta: the 13-step recursion far from the identity
g = r.gauge().copy()
r.lib.tmref_update_gauge.argtypes = [C.c_double, C.c_void_p]
for step in (0.0371, -0.2):
    r.lib.tmref_update_gauge(step, mom.ctypes.data_as(C.c_void_p))
    o.update_gauge(g, mom, step)
    assert np.array_equal(g, r.gauge()), "update_gauge step %%g" %% step
print("OK")
'''


@pytest.mark.skipif(not refbind.ref_available(hs=True), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("dims", [(4, 4, 4, 4), (4, 6, 4, 8)])
def test_oracle_update_gauge_matches_reference_bit_for_bit(dims):
    """update_gauge.c:51-110 + expo.c (exposu3, restoresu3), compiled in place into oracle/_ref/libtmref_hs.so, vs tmo_update_gauge."""
    r = subprocess.run([sys.executable, "-c", MD_CHILD % {"root": ROOT, "dims": repr(dims)}], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def test_oracle_update_gauge_reproduces_the_committed_fixture():
    import numpy as np
    from oracle.oraclebind import Oracle
    gold = os.path.join(ROOT, "tests", "golden")
    f, m = np.load(os.path.join(gold, "ref_fields_4x4.npz")), np.load(os.path.join(gold, "ref_md_4x4.npz"))
    g = np.ascontiguousarray(f["gauge"]).copy()
    Oracle(4, 4, 4, 4).update_gauge(g, np.ascontiguousarray(m["momenta"]), float(m["step"]))
    assert np.array_equal(g, m["gauge_after"])
