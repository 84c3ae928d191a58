"""TEST INFRASTRUCTURE ONLY (see oracle/README.md).

ctypes binding of oracle/_ref/libtmref*.so = the reference's own hot-path object
code (built in place from /root/reference by oracle/Makefile, never copied).
Used (a) to pin oracle/tm_oracle.c, (b) to generate tests/golden/*.npz,
(c) as bench.py's `cpu_baseline` with kind "reference".

One lattice per process: the reference keeps its state in C globals
(global.h:66-260), so `RefLattice` may be constructed once per library.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MATRIX_MULT = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


def ref_path(omp=False, hs=False):
    """omp: -O3 OpenMP build (timed CPU baseline); hs: the default half-spinor configuration with the fp32 twins."""
    name = "libtmref_hs.so" if hs else ("libtmref_omp.so" if omp else "libtmref.so")
    return os.path.join(_HERE, "_ref", name)


def ref_available(omp=False, hs=False):
    return os.path.exists(ref_path(omp, hs))


class RefLattice:
    def __init__(self, T, LX, LY, LZ, kappa=0.125, mu=0.0, nfields=12, omp=False, threads=1, hs=False):
        self.lib = lib = C.CDLL(ref_path(omp, hs))
        self.hs = hs
        self.T, self.LX, self.LY, self.LZ = T, LX, LY, LZ
        self.V = T * LX * LY * LZ
        self.nfields = nfields
        lib.tmref_init.argtypes = [C.c_int] * 4 + [C.c_double] * 2 + [C.c_int] * 2
        lib.tmref_gauge.restype = C.c_void_p
        lib.tmref_spinor.restype = C.c_void_p
        lib.tmref_spinor.argtypes = [C.c_int]
        lib.tmref_hi.restype = C.c_void_p
        lib.tmref_eo2lexic.restype = C.c_void_p
        lib.tmref_lexic2eosub.restype = C.c_void_p
        lib.tmref_set_theta.argtypes = [C.c_double] * 4
        lib.tmref_set_kappa_mu.argtypes = [C.c_double] * 2
        for name in ("Hopping_Matrix", "Hopping_Matrix_nocom"):
            getattr(lib, name).argtypes = [C.c_int, C.c_void_p, C.c_void_p]
            getattr(lib, name).restype = None
        lib.square_norm.restype = C.c_double
        lib.square_norm.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.scalar_prod_r.restype = C.c_double
        lib.scalar_prod_r.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.assign_add_mul_r.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        lib.assign_mul_add_r.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int]
        lib.assign_mul_add_r_and_square.restype = C.c_double
        lib.assign_mul_add_r_and_square.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int]
        lib.diff.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.assign.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(lib, "mul_r"):
            lib.add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            lib.mul_r.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int]
        lib.gamma5.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for name in ("Qtm_pm_psi", "Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi",
                     "D_psi", "Q_pm_psi", "Q_plus_psi", "Q_minus_psi", "Qtm_plus_sym_psi", "Qtm_minus_sym_psi",
                     "Mtm_plus_sym_psi", "Mtm_minus_sym_psi", "Mtm_plus_sym_dagg_psi", "Qtm_pm_sym_psi"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
            getattr(lib, name).restype = None
        lib.M_full.argtypes = [C.c_void_p] * 4
        lib.H_eo_tm_inv_psi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double]
        for name in ("mul_one_pm_imu_inv",):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_double, C.c_int]
        for name in ("assign_mul_one_pm_imu_inv", "assign_mul_one_pm_imu"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        lib.mul_one_pm_imu_sub_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        lib.mul_one_pm_imu_sub_mul_gamma5.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        # complex double by value: SysV passes (re, im) as two consecutive doubles in SSE regs
        lib.tm_times_Hopping_Matrix.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        lib.tm_sub_Hopping_Matrix.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        lib.cg_her.restype = C.c_int
        lib.cg_her.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
        if hs:  # fp32 twins (operator/Hopping_Matrix_32.c, operator/tm_operators_32.c, linalg/*_32.c)
            for name in ("Hopping_Matrix_32",):
                getattr(lib, name).argtypes = [C.c_int, C.c_void_p, C.c_void_p]
                getattr(lib, name).restype = None
            lib.Qtm_pm_psi_32.argtypes = [C.c_void_p, C.c_void_p]
            lib.Qtm_pm_psi_32.restype = None
            lib.assign_to_32.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            lib.assign_to_64.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            lib.square_norm_32.restype = C.c_float
            lib.square_norm_32.argtypes = [C.c_void_p, C.c_int, C.c_int]
            lib.scalar_prod_r_32.restype = C.c_float
            lib.scalar_prod_r_32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
            lib.assign_add_mul_r_32.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
            lib.assign_mul_add_r_32.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_int]
            lib.tmref_rg_mixed_cg_her.restype = C.c_int
            lib.tmref_rg_mixed_cg_her.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_int,
                                                  C.c_int, C.c_int]
        rc = lib.tmref_init(T, LX, LY, LZ, kappa, mu, nfields, threads)
        if rc != 0:
            raise RuntimeError("tmref_init failed: %d" % rc)
        self.threads = lib.tmref_threads()

    def deriv_Sb(self, ieo, il, ik, factor):
        """deriv_Sb.c:401 on g_spinor_field[il] (left) and [ik] (right), accumulating into the harness's derivative field."""
        self.lib.tmref_deriv_Sb.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double]
        self.lib.tmref_deriv_Sb(ieo, self.sp(il), self.sp(ik), factor)

    def derivative(self):
        """View of the accumulated derivative field, su3adj [V][4] as float64 [V][4][8]."""
        self.lib.tmref_derivative.restype = C.c_void_p
        n = self.V * 4 * 8
        return np.frombuffer((C.c_double * n).from_address(self.lib.tmref_derivative()), dtype=np.float64).reshape(self.V, 4, 8)

    def swpm(self):
        """Views (swm, swp) of the clover-force accumulators, su3 [V][4] as float64 [V][4][3][3][2] (clover_leaf.c:141-172)."""
        self.lib.tmref_swpm.restype = C.c_void_p
        self.lib.tmref_swpm.argtypes = [C.c_int]
        n = self.V * 4 * 18
        mk = lambda w: np.frombuffer((C.c_double * n).from_address(self.lib.tmref_swpm(w)), dtype=np.float64).reshape(self.V, 4, 3, 3, 2)
        return mk(0), mk(1)

    def rg_mixed_cg_her(self, iP, iQ, delta, max_iter, eps_sq, rel_prec, debug=0):
        """solver/rg_mixed_cg_her.c:180 on g_spinor_field[iP], [iQ] (half-spinor build only)."""
        return self.lib.tmref_rg_mixed_cg_her(self.sp(iP), self.sp(iQ), delta, max_iter, eps_sq, rel_prec, self.V // 2, debug)

    # ---- raw views onto the reference's own arrays (no copies) ----
    def gauge(self):
        """g_gauge_field as float64 [V][4][3][3][2] (lexicographic site, mu=t,x,y,z; su3.h:40-43)."""
        n = self.V * 4 * 18
        buf = (C.c_double * n).from_address(self.lib.tmref_gauge())
        return np.frombuffer(buf, dtype=np.float64).reshape(self.V, 4, 3, 3, 2)

    def spinor(self, i, nsites=None):
        """g_spinor_field[i] as float64 [nsites][4][3][2] (su3.h:60-63)."""
        nsites = nsites or self.V
        buf = (C.c_double * (nsites * 24)).from_address(self.lib.tmref_spinor(i))
        return np.frombuffer(buf, dtype=np.float64).reshape(nsites, 4, 3, 2)

    def sp(self, i):
        return self.lib.tmref_spinor(i)

    def hi(self):
        buf = (C.c_int * (16 * self.V)).from_address(self.lib.tmref_hi())
        return np.frombuffer(buf, dtype=np.int32).reshape(self.V, 16)

    def eo2lexic(self):
        buf = (C.c_int * self.V).from_address(self.lib.tmref_eo2lexic())
        return np.frombuffer(buf, dtype=np.int32)

    def random_fields(self, seed=123456):
        self.lib.tmref_random_fields(seed)
        if self.hs:
            self.lib.tmref_convert_gauge_32()

    def spinor32(self, i, nsites=None):
        """g_spinor_field32[i] as float32 [nsites][4][3][2] (su3.h:65-68); fields 0,1 are Qtm_pm_psi_32's scratch."""
        nsites = nsites or self.V // 2
        p = C.cast(C.c_void_p.in_dll(self.lib, "g_spinor_field32"), C.POINTER(C.c_void_p))[i]
        buf = (C.c_float * (nsites * 24)).from_address(p)
        return np.frombuffer(buf, dtype=np.float32).reshape(nsites, 4, 3, 2)

    def sp32(self, i):
        return C.cast(C.c_void_p.in_dll(self.lib, "g_spinor_field32"), C.POINTER(C.c_void_p))[i]

    def set_kappa_mu(self, kappa, mu):
        self.lib.tmref_set_kappa_mu(kappa, mu)

    def set_theta(self, x0, x1, x2, x3):
        self.lib.tmref_set_theta(x0, x1, x2, x3)

    def mark_gauge_dirty(self):
        self.lib.tmref_mark_gauge_dirty()

    def clover(self, c_sw, mu):
        """init_sw_fields + sw_term + sw_invert(EE, mu) (operator.c:329-330,364); returns views (sw, sw_inv)."""
        lib = self.lib
        lib.tmref_clover.argtypes = [C.c_double, C.c_double]
        lib.tmref_sw.restype = C.c_void_p
        lib.tmref_sw_inv.restype = C.c_void_p
        for n in ("Qsw_pm_psi", "Msw_plus_psi"):
            getattr(lib, n).argtypes = [C.c_void_p, C.c_void_p]
            getattr(lib, n).restype = None
        lib.clover_inv.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.clover_gamma5.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        lib.tmref_clover(c_sw, mu)
        V = self.V
        sw = np.frombuffer((C.c_double * (V * 6 * 18)).from_address(lib.tmref_sw()), dtype=np.float64).reshape(V, 3, 2, 3, 3, 2)
        swi = np.frombuffer((C.c_double * (V * 8 * 18)).from_address(lib.tmref_sw_inv()), dtype=np.float64).reshape(V, 4, 2, 3, 3, 2)
        return sw, swi

    def fnptr(self, name):
        return C.cast(getattr(self.lib, name), C.c_void_p)
