"""TEST INFRASTRUCTURE ONLY (see oracle/README.md).

ctypes binding of oracle/libtmoracle.so (oracle/tm_oracle.c, our CPU restatement
of the reference algorithm).  numpy arrays in the reference's AoS layouts:
  spinor field  float64 [nsites][4][3][2]   (su3.h:60-63)
  gauge field   float64 [VPR][4][3][3][2]   (su3.h:40-43; lexicographic site, mu = t,x,y,z)
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MATRIX_MULT = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)


def oracle_path():
    return os.path.join(_HERE, "libtmoracle.so")


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(oracle_path())
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        lib.tmo_create.restype = vp
        lib.tmo_create.argtypes = [i] * 6
        lib.tmo_destroy.argtypes = [vp]
        lib.tmo_set_threads.argtypes = [i]
        lib.tmo_get_threads.restype = i
        lib.tmo_boundary.argtypes = [vp, d, vp]
        lib.tmo_set_mu.argtypes = [vp, d]
        lib.tmo_set_mu3.argtypes = [vp, d]
        lib.tmo_set_gauge.argtypes = [vp, vp]
        lib.tmo_index.restype = i
        lib.tmo_index.argtypes = [vp, i, i, i, i]
        lib.tmo_Hopping_Matrix.argtypes = [vp, i, vp, vp]
        lib.tmo_tm_times_Hopping_Matrix.argtypes = [vp, i, vp, vp, d, d]
        lib.tmo_tm_sub_Hopping_Matrix.argtypes = [vp, i, vp, vp, vp, d, d]
        lib.tmo_D_psi.argtypes = [vp, vp, vp]
        lib.tmo_mul_one_pm_imu_inv.argtypes = [vp, vp, d, i]
        lib.tmo_assign_mul_one_pm_imu_inv.argtypes = [vp, vp, vp, d, i]
        lib.tmo_assign_mul_one_pm_imu.argtypes = [vp, vp, vp, d, i]
        lib.tmo_mul_one_pm_imu_sub_mul.argtypes = [vp, vp, vp, vp, d, i]
        lib.tmo_mul_one_pm_imu_sub_mul_gamma5.argtypes = [vp, vp, vp, vp, d]
        lib.tmo_gamma5.argtypes = [vp, vp, i]
        lib.tmo_H_eo_tm_inv_psi.argtypes = [vp, vp, vp, i, d]
        for n in ("Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi", "Qtm_pm_psi", "Qtm_plus_sym_psi",
                  "Qtm_minus_sym_psi", "Mtm_plus_sym_psi", "Mtm_minus_sym_psi", "Mtm_plus_sym_dagg_psi", "Qtm_pm_sym_psi"):
            getattr(lib, "tmo_" + n).argtypes = [vp, vp, vp]
        lib.tmo_M_full.argtypes = [vp] * 5
        lib.tmo_set_clover.argtypes = [vp, vp, vp]
        lib.tmo_sw_term.argtypes = [vp, vp, d, d]
        lib.tmo_deriv_Sb.argtypes = [vp, i, vp, vp, vp, d]
        lib.tmo_sw_spinor_eo.argtypes = [vp, i, vp, vp, vp, vp, d]
        lib.tmo_sw_deriv.argtypes = [vp, i, vp, vp, d]
        lib.tmo_sw_all.argtypes = [vp, vp, vp, vp, d, d]
        lib.tmo_update_gauge.argtypes = [vp, vp, i, d]
        lib.tmo_update_momenta.argtypes = [vp, vp, i, d]
        lib.tmo_sw_invert.argtypes = [vp, vp, vp, i, d]
        lib.tmo_sw_invert.restype = i
        lib.tmo_clover_inv.argtypes = [vp, vp, i, d]
        lib.tmo_clover_gamma5.argtypes = [vp, i, vp, vp, vp, d]
        lib.tmo_clover.argtypes = [vp, i, vp, vp, vp, d]
        lib.tmo_Qsw_pm_psi.argtypes = [vp, vp, vp]
        lib.tmo_Msw_plus_psi.argtypes = [vp, vp, vp]
        for n in ("Qsw_psi", "Qsw_plus_psi", "Qsw_minus_psi", "Qsw_sq_psi", "Msw_psi", "Msw_minus_psi"):
            getattr(lib, "tmo_" + n).argtypes = [vp, vp, vp]
        lib.tmo_assign_mul_one_sw_pm_imu.argtypes = [vp, i, vp, vp, d]
        lib.tmo_assign_mul_one_sw_pm_imu_inv.argtypes = [vp, i, vp, vp, d]
        lib.tmo_Msw_full.argtypes = [vp] * 5
        lib.tmo_square_norm.restype = d
        lib.tmo_square_norm.argtypes = [vp, i]
        lib.tmo_scalar_prod_r.restype = d
        lib.tmo_scalar_prod_r.argtypes = [vp, vp, i]
        lib.tmo_assign_add_mul_r.argtypes = [vp, vp, d, i]
        lib.tmo_assign_mul_add_r.argtypes = [vp, d, vp, i]
        lib.tmo_assign_mul_add_r_and_square.restype = d
        lib.tmo_assign_mul_add_r_and_square.argtypes = [vp, d, vp, i]
        lib.tmo_diff.argtypes = [vp, vp, vp, i]
        lib.tmo_add.argtypes = [vp, vp, vp, i]
        lib.tmo_mul_r.argtypes = [vp, C.c_double, vp, i]
        lib.tmo_assign.argtypes = [vp, vp, i]
        lib.tmo_cg_her.restype = i
        lib.tmo_cg_her.argtypes = [vp, vp, vp, i, d, i, i, vp, vp, i]
        _LIB = lib
    return _LIB


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class _LatStruct(C.Structure):
    _fields_ = [("T", C.c_int), ("LX", C.c_int), ("LY", C.c_int), ("LZ", C.c_int),
                ("nproc_t", C.c_int), ("proc_t", C.c_int),
                ("V", C.c_int), ("RAND", C.c_int), ("VPR", C.c_int),
                ("iup", C.POINTER(C.c_int)), ("idn", C.POINTER(C.c_int)),
                ("lexic2eo", C.POINTER(C.c_int)), ("lexic2eosub", C.POINTER(C.c_int)),
                ("eo2lexic", C.POINTER(C.c_int)), ("hi", C.POINTER(C.c_int))]


class Oracle:
    """One lattice (optionally one T-slab of a T-split lattice) of the CPU oracle."""

    def __init__(self, T, LX, LY, LZ, kappa=0.125, mu=0.0, theta=(0, 0, 0, 0), nproc_t=1, proc_t=0, threads=1):
        self.lib = _lib()
        self.lib.tmo_set_threads(threads)
        self.h = C.c_void_p(self.lib.tmo_create(T, LX, LY, LZ, nproc_t, proc_t))
        self.T, self.LX, self.LY, self.LZ = T, LX, LY, LZ
        st = _LatStruct.from_address(self.h.value)
        self.V, self.RAND, self.VPR = st.V, st.RAND, st.VPR
        self.Vh = self.V // 2
        self._st = st
        self.set_kappa_theta(kappa, theta)
        self.set_mu(mu)
        self._gauge = None

    def __del__(self):
        try:
            self.lib.tmo_destroy(self.h)
        except Exception:
            pass

    def table(self, name, n):
        return np.ctypeslib.as_array(getattr(self._st, name), shape=(n,))

    def eo2lexic(self):
        return self.table("eo2lexic", self.VPR)

    def lexic2eosub(self):
        return self.table("lexic2eosub", self.VPR)

    def hi(self):
        return self.table("hi", 16 * self.VPR).reshape(self.VPR, 16)

    def iup(self):
        return self.table("iup", 4 * self.VPR).reshape(self.VPR, 4)

    def idn(self):
        return self.table("idn", 4 * self.VPR).reshape(self.VPR, 4)

    def set_kappa_theta(self, kappa, theta=(0, 0, 0, 0)):
        th = (C.c_double * 4)(*theta)
        self.lib.tmo_boundary(self.h, kappa, C.cast(th, C.c_void_p))

    def set_mu(self, mu):
        self.mu = mu
        self.lib.tmo_set_mu(self.h, mu)

    def set_mu3(self, mu3):
        """g_mu3: enters the odd-odd clover term of Qsw_plus/minus/pm_psi and Msw_plus/minus_psi as g_mu + g_mu3."""
        self.lib.tmo_set_mu3(self.h, mu3)

    def set_gauge(self, g):
        assert g.shape == (self.VPR, 4, 3, 3, 2)
        self._gauge = np.ascontiguousarray(g, dtype=np.float64)
        self.lib.tmo_set_gauge(self.h, _p(self._gauge))

    def set_clover(self, sw, sw_inv):
        """sw [V][3][2][3][3][2], sw_inv [V][4][2][3][3][2] as the reference's sw_term / sw_invert lay them out."""
        assert sw.shape == (self.V, 3, 2, 3, 3, 2) and sw_inv.shape == (self.V, 4, 2, 3, 3, 2)
        self._sw = np.ascontiguousarray(sw, dtype=np.float64)
        self._sw_inv = np.ascontiguousarray(sw_inv, dtype=np.float64)
        self.lib.tmo_set_clover(self.h, _p(self._sw), _p(self._sw_inv))

    def deriv_Sb(self, ieo, l, k, df, factor):
        """deriv_Sb.c:401: accumulates the hopping part of the fermion force into df [VPR][4][8] (su3adj)."""
        assert df.shape == (self.VPR, 4, 8) and df.flags.c_contiguous
        self.lib.tmo_deriv_Sb(self.h, ieo, _p(l), _p(k), _p(df), factor)

    def sw_spinor_eo(self, ieo, swm, swp, kk, ll, fac):
        """operator/clover_deriv.c:252; swm / swp: [V][4][3][3][2], accumulated."""
        self.lib.tmo_sw_spinor_eo(self.h, ieo, _p(swm), _p(swp), _p(kk), _p(ll), fac)

    def sw_deriv(self, ieo, swm, swp, mu):
        self.lib.tmo_sw_deriv(self.h, ieo, _p(swm), _p(swp), mu)

    def sw_all(self, df, swm, swp, kappa, c_sw):
        assert df.shape == (self.VPR, 4, 8) and df.flags.c_contiguous
        self.lib.tmo_sw_all(self.h, _p(df), _p(swm), _p(swp), kappa, c_sw)

    def update_gauge(self, gauge, mom, step):
        """update_gauge.c:51-110 in place on gauge [V][4][3][3][2] with momenta [V][4][8] (both host arrays; V from the momenta)."""
        self.lib.tmo_update_gauge(_p(gauge), _p(mom), mom.shape[0], step)

    def update_momenta(self, mom, deriv, step):
        self.lib.tmo_update_momenta(_p(mom), _p(deriv), mom.shape[0], step)

    def sw_term(self, kappa, c_sw):
        """operator/clover_term.c:88 on the current gauge field -> sw [V][3][2][3][3][2]."""
        sw = np.zeros((self.V, 3, 2, 3, 3, 2))
        self.lib.tmo_sw_term(self.h, _p(sw), kappa, c_sw)
        return sw

    def sw_invert(self, sw, ieo, mu):
        """operator/clover_invert.c:170 -> (sw_inv [V][4][2][3][3][2], number of near-singular pivots)."""
        swi = np.zeros((self.V, 4, 2, 3, 3, 2))
        sw = np.ascontiguousarray(sw, dtype=np.float64)
        fails = self.lib.tmo_sw_invert(self.h, _p(swi), _p(sw), ieo, mu)
        return swi, fails

    def assign_mul_one_sw_pm_imu(self, ieo, k, l, mu):
        self.lib.tmo_assign_mul_one_sw_pm_imu(self.h, ieo, _p(k), _p(l), mu)

    def assign_mul_one_sw_pm_imu_inv(self, ieo, k, l, mu):
        self.lib.tmo_assign_mul_one_sw_pm_imu_inv(self.h, ieo, _p(k), _p(l), mu)

    def Msw_full(self, en, on, e, o):
        self.lib.tmo_Msw_full(self.h, _p(en), _p(on), _p(e), _p(o))

    def clover_inv(self, l, tau3sign, mu):
        self.lib.tmo_clover_inv(self.h, _p(l), tau3sign, mu)

    def clover_gamma5(self, ieo, l, k, j, mu):
        self.lib.tmo_clover_gamma5(self.h, ieo, _p(l), _p(k), _p(j), mu)

    def clover(self, ieo, l, k, j, mu):
        self.lib.tmo_clover(self.h, ieo, _p(l), _p(k), _p(j), mu)

    def new_field(self, n=None):
        return np.zeros((n or (self.VPR // 2), 4, 3, 2), dtype=np.float64)

    # --- operators (names follow the reference) ---
    def Hopping_Matrix(self, ieo, l, k):
        self.lib.tmo_Hopping_Matrix(self.h, ieo, _p(l), _p(k))

    def tm_times_Hopping_Matrix(self, ieo, l, k, c):
        self.lib.tmo_tm_times_Hopping_Matrix(self.h, ieo, _p(l), _p(k), c.real, c.imag)

    def tm_sub_Hopping_Matrix(self, ieo, l, p, k, c):
        self.lib.tmo_tm_sub_Hopping_Matrix(self.h, ieo, _p(l), _p(p), _p(k), c.real, c.imag)

    def D_psi(self, P, Q):
        self.lib.tmo_D_psi(self.h, _p(P), _p(Q))

    def mul_one_pm_imu_inv(self, l, sign, N):
        self.lib.tmo_mul_one_pm_imu_inv(self.h, _p(l), sign, N)

    def assign_mul_one_pm_imu_inv(self, l, k, sign, N):
        self.lib.tmo_assign_mul_one_pm_imu_inv(self.h, _p(l), _p(k), sign, N)

    def assign_mul_one_pm_imu(self, l, k, sign, N):
        self.lib.tmo_assign_mul_one_pm_imu(self.h, _p(l), _p(k), sign, N)

    def mul_one_pm_imu_sub_mul(self, l, k, j, sign, N):
        self.lib.tmo_mul_one_pm_imu_sub_mul(self.h, _p(l), _p(k), _p(j), sign, N)

    def mul_one_pm_imu_sub_mul_gamma5(self, l, k, j, sign):
        self.lib.tmo_mul_one_pm_imu_sub_mul_gamma5(self.h, _p(l), _p(k), _p(j), sign)

    def gamma5(self, l, k, N):
        self.lib.tmo_gamma5(_p(l), _p(k), N)

    def H_eo_tm_inv_psi(self, l, k, ieo, sign):
        self.lib.tmo_H_eo_tm_inv_psi(self.h, _p(l), _p(k), ieo, sign)

    def op(self, name, l, k):
        getattr(self.lib, "tmo_" + name)(self.h, _p(l), _p(k))

    def M_full(self, en, on, e, o):
        self.lib.tmo_M_full(self.h, _p(en), _p(on), _p(e), _p(o))

    # --- linalg ---
    def square_norm(self, P, N):
        return self.lib.tmo_square_norm(_p(P), N)

    def scalar_prod_r(self, S, R, N):
        return self.lib.tmo_scalar_prod_r(_p(S), _p(R), N)

    def assign_add_mul_r(self, P, Q, c, N):
        self.lib.tmo_assign_add_mul_r(_p(P), _p(Q), c, N)

    def assign_mul_add_r(self, R, c, S, N):
        self.lib.tmo_assign_mul_add_r(_p(R), c, _p(S), N)

    def assign_mul_add_r_and_square(self, R, c, S, N):
        return self.lib.tmo_assign_mul_add_r_and_square(_p(R), c, _p(S), N)

    def diff(self, Q, R, S, N):
        self.lib.tmo_diff(_p(Q), _p(R), _p(S), N)

    def add(self, Q, R, S, N):
        self.lib.tmo_add(_p(Q), _p(R), _p(S), N)

    def mul_r(self, R, c, S, N):
        self.lib.tmo_mul_r(_p(R), c, _p(S), N)

    def cg_her(self, P, Q, max_iter, eps_sq, rel_prec, N, opname="Qtm_pm_psi"):
        f = C.cast(getattr(self.lib, "tmo_" + opname), C.c_void_p)
        hist = np.zeros(max_iter, dtype=np.float64)
        it = self.lib.tmo_cg_her(self.h, _p(P), _p(Q), max_iter, eps_sq, rel_prec, N, f, _p(hist), max_iter)
        return it, hist[: max(it, 0)] if it > 0 else hist
