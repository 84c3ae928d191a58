"""ctypes mirror of include/tmlqcd_hip.h (host-side plumbing only; no numerics here).

Function names follow the reference (operator/tm_operators.h, linalg/*.h, solver/cg_her.h);
host arrays are numpy float64 in the reference's AoS layouts:
  spinor field [nsites][4][3][2]  (su3.h:60-63),  gauge field [VPR][4][3][3][2]  (su3.h:40-43).
"""
import ctypes as C
import os
import numpy as np

EO, OE = 0, 1
FIELD_EO, FIELD_FULL = 0, 1
OPS = {"Qtm_pm_psi": 0, "Qtm_plus_psi": 1, "Qtm_minus_psi": 2, "Mtm_plus_psi": 3, "Mtm_minus_psi": 4, "Qsw_pm_psi": 5}
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class TmHipError(RuntimeError):
    pass


def library_path():
    # TMLQCD_HIP_LIB: another build of the same library (A/B runs of kernel variants, tools/)
    return os.environ.get("TMLQCD_HIP_LIB") or os.path.join(_HERE, "lib", "libtmlqcd_hip.so")


class GaugeInfo(C.Structure):
    """tmhip_gauge_info (include/tmlqcd_hip.h): what the reference keeps in GaugeInfo after read_gauge_field."""
    _fields_ = [("gauge_read", C.c_int), ("suma", C.c_uint), ("sumb", C.c_uint), ("suma_stored", C.c_uint), ("sumb_stored", C.c_uint),
                ("prec", C.c_int), ("lx", C.c_int), ("ly", C.c_int), ("lz", C.c_int), ("lt", C.c_int),
                ("xlf_info", C.c_char * 1024), ("ildg_data_lfn", C.c_char * 512)]


class _Geom(C.Structure):
    _fields_ = [("T", C.c_int), ("LX", C.c_int), ("LY", C.c_int), ("LZ", C.c_int),
                ("nproc_t", C.c_int), ("proc_t", C.c_int)]


def load_library():
    """Load libtmlqcd_hip.so.  Fails loudly: there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise TmHipError("%s not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(or make -C tmlqcd_amd/csrc); there is no CPU fallback" % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    pd = C.POINTER(C.c_double)
    sig = {
        "tmhip_create": [C.POINTER(_Geom), i, C.POINTER(vp)],
        "tmhip_sync": [vp],
        "tmhip_set_boundary": [vp, d, pd],
        "tmhip_set_mu": [vp, d],
        "tmhip_set_mu3": [vp, d],
        "tmhip_set_gauge": [vp, vp],
        "tmhip_field_alloc": [vp, i, C.POINTER(vp)],
        "tmhip_field_upload": [vp, vp, vp, i],
        "tmhip_field_download": [vp, vp, vp, i],
        "tmhip_field_zero": [vp, vp],
        "tmhip_hopping_matrix": [vp, i, vp, vp],
        "tmhip_hopping_matrix_nocom": [vp, i, vp, vp],
        "tmhip_tm_times_hopping_matrix": [vp, i, vp, vp, d, d],
        "tmhip_tm_sub_hopping_matrix": [vp, i, vp, vp, vp, d, d],
        "tmhip_D_psi": [vp, vp, vp],
        "tmhip_mul_one_pm_imu_inv": [vp, vp, d, i],
        "tmhip_assign_mul_one_pm_imu_inv": [vp, vp, vp, d, i],
        "tmhip_assign_mul_one_pm_imu": [vp, vp, vp, d, i],
        "tmhip_mul_one_pm_imu": [vp, vp, d],
        "tmhip_mul_one_pm_imu_sub_mul": [vp, vp, vp, vp, d, i],
        "tmhip_mul_one_pm_imu_sub_mul_gamma5": [vp, vp, vp, vp, d],
        "tmhip_gamma5": [vp, vp, vp, i],
        "tmhip_H_eo_tm_inv_psi": [vp, vp, vp, i, d],
        "tmhip_Qtm_plus_psi": [vp, vp, vp], "tmhip_Qtm_minus_psi": [vp, vp, vp],
        "tmhip_Mtm_plus_psi": [vp, vp, vp], "tmhip_Mtm_minus_psi": [vp, vp, vp],
        "tmhip_Qtm_pm_psi": [vp, vp, vp],
        "tmhip_Qtm_plus_sym_psi": [vp, vp, vp], "tmhip_Qtm_minus_sym_psi": [vp, vp, vp],
        "tmhip_Mtm_plus_sym_psi": [vp, vp, vp], "tmhip_Mtm_minus_sym_psi": [vp, vp, vp],
        "tmhip_Mtm_plus_sym_dagg_psi": [vp, vp, vp], "tmhip_Qtm_pm_sym_psi": [vp, vp, vp],
        "tmhip_mul_one_sub_mul_gamma5": [vp, vp, vp, vp],
        "tmhip_M_full": [vp, vp, vp, vp, vp],
        "tmhip_square_norm": [vp, vp, i, i, pd],
        "tmhip_scalar_prod_r": [vp, vp, vp, i, i, pd],
        "tmhip_assign_add_mul_r": [vp, vp, vp, d, i],
        "tmhip_assign_mul_add_r": [vp, vp, d, vp, i],
        "tmhip_assign_mul_add_r_and_square": [vp, vp, d, vp, i, i, pd],
        "tmhip_diff": [vp, vp, vp, vp, i],
        "tmhip_assign": [vp, vp, vp, i],
        "tmhip_add": [vp, vp, vp, vp, i],
        "tmhip_mul_r": [vp, vp, d, vp, i],
        "tmhip_cg_her": [vp, vp, vp, i, d, i, i, i, C.POINTER(i), pd, i],
        "tmhip_set_clover": [vp, vp, vp],
        "tmhip_sw_term": [vp, vp, d, d],
        "tmhip_momenta_upload": [vp, vp], "tmhip_momenta_download": [vp, vp], "tmhip_update_momenta": [vp, d],
        "tmhip_update_gauge": [vp, d], "tmhip_gauge_download": [vp, vp],
        "tmhip_gauge_unpack_ildg": [vp, vp, i, C.POINTER(C.c_uint)], "tmhip_gauge_pack_ildg": [vp, vp, i, C.POINTER(C.c_uint)],
        "tmhip_read_gauge_field": [vp, C.c_char_p, i, i, vp, C.POINTER(GaugeInfo)],
        "tmhip_write_gauge_field": [vp, C.c_char_p, i, C.c_char_p, C.POINTER(C.c_uint)],
        "tmhip_gauge_su3_deviation": [vp, pd],
        "tmhip_derivative_zero": [vp],
        "tmhip_swpm_zero": [vp], "tmhip_sw_spinor_eo": [vp, i, vp, vp, d], "tmhip_sw_deriv": [vp, i, d],
        "tmhip_sw_all": [vp, vp, d, d], "tmhip_get_swpm": [vp, vp, vp],
        "tmhip_deriv_Sb": [vp, i, vp, vp, d],
        "tmhip_derivative_download": [vp, vp, i],
        "tmhip_multi_deriv_Sb": [i, C.POINTER(vp), i, C.POINTER(vp), C.POINTER(vp), d],
        "tmhip_multi_sw_all": [i, C.POINTER(vp), d, d],
        "tmhip_multi_update_gauge": [i, C.POINTER(vp), d],
        "tmhip_sw_invert": [vp, i, d],
        "tmhip_get_clover": [vp, vp, vp],
        "tmhip_clover_inv": [vp, vp, i, d],
        "tmhip_clover_gamma5": [vp, i, vp, vp, vp, d],
        "tmhip_clover": [vp, i, vp, vp, vp, d],
        "tmhip_H_eo_sw_inv_psi": [vp, vp, vp, i, i, d],
        "tmhip_Qsw_pm_psi": [vp, vp, vp],
        "tmhip_Msw_plus_psi": [vp, vp, vp],
        "tmhip_Qsw_psi": [vp, vp, vp], "tmhip_Qsw_plus_psi": [vp, vp, vp], "tmhip_Qsw_minus_psi": [vp, vp, vp],
        "tmhip_Qsw_sq_psi": [vp, vp, vp], "tmhip_Msw_psi": [vp, vp, vp], "tmhip_Msw_minus_psi": [vp, vp, vp],
        "tmhip_Msw_full": [vp, vp, vp, vp, vp],
        "tmhip_assign_mul_one_sw_pm_imu": [vp, i, vp, vp, d], "tmhip_assign_mul_one_sw_pm_imu_inv": [vp, i, vp, vp, d],
        "tmhip_Qsw_pm_psi_32": [vp, vp, vp],
        "tmhip_field_alloc32": [vp, C.POINTER(vp)],
        "tmhip_field_upload32": [vp, vp, vp, i],
        "tmhip_field_download32": [vp, vp, vp, i],
        "tmhip_assign_to_32": [vp, vp, vp, i],
        "tmhip_assign_to_64": [vp, vp, vp, i],
        "tmhip_add_from_32": [vp, vp, vp, i],
        "tmhip_hopping_matrix_32": [vp, i, vp, vp],
        "tmhip_Qtm_pm_psi_32": [vp, vp, vp],
        "tmhip_square_norm_32": [vp, vp, i, i, pd],
        "tmhip_scalar_prod_r_32": [vp, vp, vp, i, i, pd],
        "tmhip_assign_add_mul_r_32": [vp, vp, vp, C.c_float, i],
        "tmhip_assign_mul_add_r_32": [vp, vp, C.c_float, vp, i],
        "tmhip_mixed_cg_her": [vp, vp, vp, i, d, i, i, i, d, i, C.POINTER(i), C.POINTER(i)],
        "tmhip_mixed_cg_restarts": [vp, C.POINTER(i), i, C.POINTER(i)],
        "tmhip_rg_mixed_cg_her": [vp, vp, vp, i, d, i, i, i, d, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)],
        "tmhip_comm_get_unique_id": [C.c_char_p],
        "tmhip_comm_init": [vp, C.c_char_p],
        "tmhip_comm_init_shm": [vp, C.c_char_p],
        "tmhip_comm_init_ipc": [vp],
        "tmhip_comm_faces_direct": [vp, C.POINTER(i)],
        "tmhip_comm_sums_direct": [vp],
        "tmhip_comm_set_loopback": [vp, i],
        "tmhip_comm_count": [vp, C.POINTER(i), C.POINTER(i)],
        "tmhip_comm_is_split": [vp],
        "tmhip_comm_stream_delay_ms": [vp, i],
        "tmhip_bench_hopping": [vp, vp, vp, vp, i, pd],
        "tmhip_multi_hopping_matrix": [i, C.POINTER(vp), i, C.POINTER(vp), C.POINTER(vp)],
        "tmhip_event_record": [vp, i],
        "tmhip_event_elapsed_ms": [vp, i, i, pd],
        "tmhip_set_option": [vp, C.c_char_p, i],
    }
    for name, args in sig.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = i
    lib.tmhip_destroy.argtypes = [vp]
    lib.tmhip_destroy.restype = None
    lib.tmhip_field_free.argtypes = [vp, vp]
    lib.tmhip_field_free.restype = None
    lib.tmhip_field_even.argtypes = [vp]
    lib.tmhip_field_even.restype = vp
    lib.tmhip_field_odd.argtypes = [vp]
    lib.tmhip_field_odd.restype = vp
    lib.tmhip_version.restype = C.c_char_p
    lib.tmhip_device_count.restype = i
    _LIB = lib
    return lib


def _ck(rc, what):
    if rc != 0:
        raise TmHipError("%s failed (rc=%d); see stderr" % (what, rc))


def _hp(a):
    if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
        raise TmHipError("host arrays must be C-contiguous float64")
    return a.ctypes.data_as(C.c_void_p)


class Field:
    """A device-resident spinor field (opaque handle); prec 32 = the reference's spinor32."""

    def __init__(self, lat, kind=FIELD_EO, handle=None, owner=True, prec=64):
        self.lat, self.kind, self.owner, self.prec = lat, kind, owner, prec
        if handle is None:
            h = C.c_void_p()
            if prec == 32:
                _ck(lat.lib.tmhip_field_alloc32(lat.h, C.byref(h)), "tmhip_field_alloc32")
            else:
                _ck(lat.lib.tmhip_field_alloc(lat.h, kind, C.byref(h)), "tmhip_field_alloc")
            handle = h
        self.h = handle

    def upload(self, host, nsites=None):
        nsites = nsites if nsites is not None else (self.lat.V if self.kind == FIELD_FULL else self.lat.Vh)
        if self.prec == 32:
            if host.dtype != np.float32 or not host.flags["C_CONTIGUOUS"]:
                raise TmHipError("fp32 fields take C-contiguous float32 arrays")
            _ck(self.lat.lib.tmhip_field_upload32(self.lat.h, self.h, host.ctypes.data_as(C.c_void_p), nsites), "tmhip_field_upload32")
            return self
        _ck(self.lat.lib.tmhip_field_upload(self.lat.h, self.h, _hp(host), nsites), "tmhip_field_upload")
        return self

    def download(self, nsites=None, out=None):
        """Returns the field as a host AoS array; `out`: an existing fp64 array of the right shape to fill instead (a host
        program's own, already touched, buffer)."""
        nsites = nsites if nsites is not None else (self.lat.V if self.kind == FIELD_FULL else self.lat.Vh)
        if out is not None:
            if self.prec == 32 or out.dtype != np.float64 or out.shape != (nsites, 4, 3, 2) or not out.flags.c_contiguous:
                raise TmHipError("download(out=...): needs a C-contiguous float64 [%d][4][3][2] array and an fp64 field" % nsites)
            _ck(self.lat.lib.tmhip_field_download(self.lat.h, self.h, _hp(out), nsites), "tmhip_field_download")
            return out
        if self.prec == 32:
            out = np.empty((nsites, 4, 3, 2), dtype=np.float32)
            _ck(self.lat.lib.tmhip_field_download32(self.lat.h, self.h, out.ctypes.data_as(C.c_void_p), nsites), "tmhip_field_download32")
            return out
        out = np.empty((nsites, 4, 3, 2), dtype=np.float64)
        _ck(self.lat.lib.tmhip_field_download(self.lat.h, self.h, _hp(out), nsites), "tmhip_field_download")
        return out

    def zero(self):
        _ck(self.lat.lib.tmhip_field_zero(self.lat.h, self.h), "tmhip_field_zero")
        return self

    def even(self):
        return Field(self.lat, FIELD_EO, C.c_void_p(self.lat.lib.tmhip_field_even(self.h)), owner=False)

    def odd(self):
        return Field(self.lat, FIELD_EO, C.c_void_p(self.lat.lib.tmhip_field_odd(self.h)), owner=False)

    def free(self):
        if self.owner and self.h is not None and self.lat.h is not None:
            self.lat.lib.tmhip_field_free(self.lat.h, self.h)
        self.h = None


class Lattice:
    """One rank's lattice on one MI355X: context + operators, mirroring the reference's names."""

    def __init__(self, T, LX, LY, LZ, kappa=0.125, mu=0.0, theta=(0, 0, 0, 0), nproc_t=1, proc_t=0, device=0):
        self.lib = load_library()
        self.T, self.LX, self.LY, self.LZ = T, LX, LY, LZ
        self.nproc_t, self.proc_t = nproc_t, proc_t
        self.V = T * LX * LY * LZ
        self.Vh = self.V // 2
        self.VPR = self.V + (2 * LX * LY * LZ if nproc_t > 1 else 0)
        g = _Geom(T, LX, LY, LZ, nproc_t, proc_t)
        h = C.c_void_p()
        self.h = None
        _ck(self.lib.tmhip_create(C.byref(g), device, C.byref(h)), "tmhip_create")
        self.h = h
        self.set_boundary(kappa, theta)
        self.set_mu(mu)

    def close(self):
        if self.h is not None:
            self.lib.tmhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- parameters -------------------------------------------------------
    def set_boundary(self, kappa, theta=(0, 0, 0, 0)):
        th = (C.c_double * 4)(*theta)
        _ck(self.lib.tmhip_set_boundary(self.h, kappa, th), "tmhip_set_boundary")

    def set_mu(self, mu):
        self.mu = mu
        _ck(self.lib.tmhip_set_mu(self.h, mu), "tmhip_set_mu")

    def set_mu3(self, mu3):
        _ck(self.lib.tmhip_set_mu3(self.h, mu3), "tmhip_set_mu3")

    def set_gauge(self, g):
        if g.shape != (self.VPR, 4, 3, 3, 2):
            raise TmHipError("gauge field must be [VPR=%d][4][3][3][2], got %s" % (self.VPR, g.shape))
        _ck(self.lib.tmhip_set_gauge(self.h, _hp(g)), "tmhip_set_gauge")

    def set_option(self, name, value):
        _ck(self.lib.tmhip_set_option(self.h, name.encode(), int(value)), "tmhip_set_option")

    def gauge_su3_deviation(self):
        """max |row2 - conj(row0 x row1)| over the resident links (what the gauge_recon=12 guard looks at)."""
        out = C.c_double()
        _ck(self.lib.tmhip_gauge_su3_deviation(self.h, C.byref(out)), "tmhip_gauge_su3_deviation")
        return out.value

    def sync(self):
        _ck(self.lib.tmhip_sync(self.h), "tmhip_sync")

    # --- fields -----------------------------------------------------------
    def field(self, host=None, kind=FIELD_EO):
        f = Field(self, kind)
        if host is not None:
            f.upload(host, host.shape[0])
        return f

    def full_field(self, host=None):
        return self.field(host, FIELD_FULL)

    def field32(self, host=None):
        f = Field(self, FIELD_EO, prec=32)
        if host is not None:
            f.upload(host, host.shape[0])
        return f

    # --- clover twisted mass (operator/clovertm_operators.h) ---------------
    def set_clover(self, sw, sw_inv):
        """sw [V][3][2][3][3][2] from sw_term, sw_inv [V][4][2][3][3][2] from sw_invert(EE, mu) (host arrays)."""
        if sw.shape != (self.V, 3, 2, 3, 3, 2) or sw_inv.shape != (self.V, 4, 2, 3, 3, 2):
            raise TmHipError("clover arrays must be [V][3][2][3][3][2] and [V][4][2][3][3][2]")
        _ck(self.lib.tmhip_set_clover(self.h, _hp(sw), _hp(sw_inv)), "tmhip_set_clover")

    # --- fermion force (deriv_Sb.h) ------------------------------------------
    def derivative_zero(self):
        _ck(self.lib.tmhip_derivative_zero(self.h), "tmhip_derivative_zero")

    def deriv_Sb(self, ieo, l, k, factor):
        """deriv_Sb.c:401: accumulate the hopping part of the fermion force into the device-resident derivative field."""
        _ck(self.lib.tmhip_deriv_Sb(self.h, ieo, l.h, k.h, factor), "deriv_Sb")

    def derivative(self, into=None):
        """The accumulated derivative as su3adj df[V][4][8]; with `into`, added to that host array (accumulate)."""
        if into is None:
            out = np.zeros((self.V, 4, 8))
            _ck(self.lib.tmhip_derivative_download(self.h, _hp(out), 0), "tmhip_derivative_download")
            return out
        _ck(self.lib.tmhip_derivative_download(self.h, _hp(into), 1), "tmhip_derivative_download")
        return into

    # --- clover part of the force (clover_deriv.c, clover_accumulate_deriv.c) ---
    def swpm_zero(self):
        _ck(self.lib.tmhip_swpm_zero(self.h), "tmhip_swpm_zero")

    def sw_spinor_eo(self, ieo, kk, ll, fac):
        _ck(self.lib.tmhip_sw_spinor_eo(self.h, ieo, kk.h, ll.h, fac), "sw_spinor_eo")

    def sw_deriv(self, ieo, mu):
        _ck(self.lib.tmhip_sw_deriv(self.h, ieo, mu), "sw_deriv")

    def sw_all(self, kappa, c_sw, gauge=None):
        _ck(self.lib.tmhip_sw_all(self.h, _hp(gauge) if gauge is not None else None, kappa, c_sw), "sw_all")

    def get_swpm(self):
        swm, swp = np.zeros((self.V, 4, 3, 3, 2)), np.zeros((self.V, 4, 3, 3, 2))
        _ck(self.lib.tmhip_get_swpm(self.h, _hp(swm), _hp(swp)), "tmhip_get_swpm")
        return swm, swp

    def sw_term(self, gauge, kappa, c_sw):
        """operator/clover_term.c:88 on the device; `gauge` as for set_gauge ([VPR][4][3][3][2]), or None = the links resident
        on the device (after set_gauge / update_gauge)."""
        if gauge is not None and gauge.shape != (self.VPR, 4, 3, 3, 2):
            raise TmHipError("gauge field must be [%d][4][3][3][2]" % self.VPR)
        _ck(self.lib.tmhip_sw_term(self.h, _hp(gauge) if gauge is not None else None, kappa, c_sw), "tmhip_sw_term")

    # --- molecular dynamics with the links resident in HBM (update_gauge.c, update_momenta.c) ---------------
    def momenta_upload(self, mom):
        if mom.shape != (self.V, 4, 8):
            raise TmHipError("momenta must be [V=%d][4][8]" % self.V)
        _ck(self.lib.tmhip_momenta_upload(self.h, _hp(mom)), "tmhip_momenta_upload")

    def momenta_download(self):
        out = np.zeros((self.V, 4, 8))
        _ck(self.lib.tmhip_momenta_download(self.h, _hp(out)), "tmhip_momenta_download")
        return out

    def update_momenta(self, step):
        _ck(self.lib.tmhip_update_momenta(self.h, step), "update_momenta")

    def update_gauge(self, step):
        _ck(self.lib.tmhip_update_gauge(self.h, step), "update_gauge")

    def gauge_download(self):
        out = np.zeros((self.VPR, 4, 3, 3, 2))
        _ck(self.lib.tmhip_gauge_download(self.h, _hp(out)), "tmhip_gauge_download")
        return out

    # ---- ILDG gauge configurations (io/gauge_read.c, io/gauge_write.c; ildg.hip)
    def gauge_unpack_ildg(self, file_bytes, prec):
        """This rank's part of an "ildg-binary-data" record (bytes / uint8 array) -> resident links; returns (suma, sumb)."""
        buf = np.frombuffer(file_bytes, dtype=np.uint8) if not isinstance(file_bytes, np.ndarray) else file_bytes
        want = self.V * 4 * 9 * (16 if prec == 64 else 8)
        if buf.size != want:
            raise TmHipError("gauge_unpack_ildg: %d bytes, expected %d" % (buf.size, want))
        buf = np.ascontiguousarray(buf)
        sums = (C.c_uint * 2)()
        _ck(self.lib.tmhip_gauge_unpack_ildg(self.h, buf.ctypes.data_as(C.c_void_p), prec, sums), "tmhip_gauge_unpack_ildg")
        return int(sums[0]), int(sums[1])

    def gauge_pack_ildg(self, prec, out=None):
        """The resident links as the bytes of this rank's part of the record; returns (uint8 array, (suma, sumb)).  `out`: a buffer
        of the record's size to fill (a fresh numpy array is 600 MB of pages the kernel has never mapped: the copy into it
        runs at the page-fault rate, 52 ms instead of 12 at 32^4)."""
        n = self.V * 4 * 9 * (16 if prec == 64 else 8)
        if out is None:
            out = np.zeros(n, dtype=np.uint8)
        elif out.dtype != np.uint8 or out.size != n or not out.flags["C_CONTIGUOUS"]:
            raise TmHipError("gauge_pack_ildg: out must be a contiguous uint8 array of %d bytes" % n)
        sums = (C.c_uint * 2)()
        _ck(self.lib.tmhip_gauge_pack_ildg(self.h, out.ctypes.data_as(C.c_void_p), prec, sums), "tmhip_gauge_pack_ildg")
        return out, (int(sums[0]), int(sums[1]))

    def read_gauge_field(self, filename, prec=64, io_checks=True, want_host=True):
        """read_gauge_field(filename, gf) (io/gauge_read.c:28): returns (status 0 | -1, host field or None, GaugeInfo)."""
        info = GaugeInfo()
        out = np.zeros((self.VPR, 4, 3, 3, 2)) if want_host else None
        rc = self.lib.tmhip_read_gauge_field(self.h, str(filename).encode(), prec, 1 if io_checks else 0, _hp(out) if want_host else None, C.byref(info))
        if rc not in (0, -1):
            raise TmHipError("tmhip_read_gauge_field failed (rc=%d)" % rc)
        return rc, out, info

    def write_gauge_field(self, filename, prec=64, xlf_info=None):
        """write_gauge_field(filename, prec, xlfInfo) (io/gauge_write.c:22) from the resident links; returns (suma, sumb)."""
        sums = (C.c_uint * 2)()
        _ck(self.lib.tmhip_write_gauge_field(self.h, str(filename).encode(), prec, xlf_info.encode() if xlf_info else None, sums), "tmhip_write_gauge_field")
        return int(sums[0]), int(sums[1])

    def sw_invert(self, ieo, mu):
        """operator/clover_invert.c:170 on the device (needs sw_term or set_clover first)."""
        _ck(self.lib.tmhip_sw_invert(self.h, ieo, mu), "tmhip_sw_invert")

    def get_clover(self, want_sw=True, want_sw_inv=True):
        """Device-resident clover blocks in the reference's host layouts: (sw [V][3][2][3][3][2], sw_inv [V][4][2][3][3][2])."""
        sw = np.zeros((self.V, 3, 2, 3, 3, 2)) if want_sw else None
        swi = np.zeros((self.V, 4, 2, 3, 3, 2)) if want_sw_inv else None
        _ck(self.lib.tmhip_get_clover(self.h, _hp(sw) if want_sw else None, _hp(swi) if want_sw_inv else None), "tmhip_get_clover")
        return sw, swi

    def assign_mul_one_sw_pm_imu(self, ieo, k, l, mu):
        _ck(self.lib.tmhip_assign_mul_one_sw_pm_imu(self.h, ieo, k.h, l.h, mu), "assign_mul_one_sw_pm_imu")

    def assign_mul_one_sw_pm_imu_inv(self, ieo, k, l, mu):
        _ck(self.lib.tmhip_assign_mul_one_sw_pm_imu_inv(self.h, ieo, k.h, l.h, mu), "assign_mul_one_sw_pm_imu_inv")

    def Msw_full(self, en, on, e, o):
        _ck(self.lib.tmhip_Msw_full(self.h, en.h, on.h, e.h, o.h), "Msw_full")

    def clover_inv(self, l, tau3sign, mu):
        _ck(self.lib.tmhip_clover_inv(self.h, l.h, tau3sign, mu), "clover_inv")

    def clover_gamma5(self, ieo, l, k, j, mu):
        _ck(self.lib.tmhip_clover_gamma5(self.h, ieo, l.h, k.h, j.h, mu), "clover_gamma5")

    def clover(self, ieo, l, k, j, mu):
        _ck(self.lib.tmhip_clover(self.h, ieo, l.h, k.h, j.h, mu), "clover")

    def H_eo_sw_inv_psi(self, l, k, ieo, tau3sign, mu):
        _ck(self.lib.tmhip_H_eo_sw_inv_psi(self.h, l.h, k.h, ieo, tau3sign, mu), "H_eo_sw_inv_psi")

    def Qsw_pm_psi_32(self, l, k):
        _ck(self.lib.tmhip_Qsw_pm_psi_32(self.h, l.h, k.h), "Qsw_pm_psi_32")

    # --- mixed precision (reference names with the _32 suffix) -------------
    def assign_to_32(self, r32, s64, N):
        _ck(self.lib.tmhip_assign_to_32(self.h, r32.h, s64.h, N), "assign_to_32")

    def assign_to_64(self, r64, s32, N):
        _ck(self.lib.tmhip_assign_to_64(self.h, r64.h, s32.h, N), "assign_to_64")

    def Hopping_Matrix_32(self, ieo, l, k):
        _ck(self.lib.tmhip_hopping_matrix_32(self.h, ieo, l.h, k.h), "Hopping_Matrix_32")

    def Qtm_pm_psi_32(self, l, k):
        _ck(self.lib.tmhip_Qtm_pm_psi_32(self.h, l.h, k.h), "Qtm_pm_psi_32")

    def square_norm_32(self, P, N, parallel=0):
        out = C.c_double()
        _ck(self.lib.tmhip_square_norm_32(self.h, P.h, N, parallel, C.byref(out)), "square_norm_32")
        return out.value

    def scalar_prod_r_32(self, S, R, N, parallel=0):
        out = C.c_double()
        _ck(self.lib.tmhip_scalar_prod_r_32(self.h, S.h, R.h, N, parallel, C.byref(out)), "scalar_prod_r_32")
        return out.value

    def assign_add_mul_r_32(self, P, Q, c, N):
        _ck(self.lib.tmhip_assign_add_mul_r_32(self.h, P.h, Q.h, c, N), "assign_add_mul_r_32")

    def assign_mul_add_r_32(self, R, c, S, N):
        _ck(self.lib.tmhip_assign_mul_add_r_32(self.h, R.h, c, S.h, N), "assign_mul_add_r_32")

    def mixed_cg_her(self, P, Q, max_iter, eps_sq, rel_prec, N, innereps=5.0e-5, max_inner_it=5000, op="Qtm_pm_psi"):
        """solver/mixed_cg_her.c with (f, f32) = (Qtm_pm_psi, Qtm_pm_psi_32) or (Qsw_pm_psi, Qsw_pm_psi_32);
        returns (iterations, outer iterations)."""
        it, outer = C.c_int(), C.c_int()
        _ck(self.lib.tmhip_mixed_cg_her(self.h, P.h, Q.h, max_iter, eps_sq, rel_prec, N, OPS[op], innereps,
                                        max_inner_it, C.byref(it), C.byref(outer)), "mixed_cg_her")
        return it.value, outer.value

    def mixed_cg_restarts(self):
        """Inner iteration count (the reference's j, mixed_cg_her.c:152) of every outer iteration of the last mixed_cg_her."""
        buf, n = (C.c_int * 256)(), C.c_int()
        _ck(self.lib.tmhip_mixed_cg_restarts(self.h, buf, 256, C.byref(n)), "tmhip_mixed_cg_restarts")
        return list(buf[:min(n.value, 256)])

    def rg_mixed_cg_her(self, P, Q, max_iter, eps_sq, rel_prec, N, delta=5.0e-5, op="Qtm_pm_psi"):
        """solver/rg_mixed_cg_her.c:180 (reliable updates, fp64 fail-safe); delta = solver_params.mcg_delta.
        Returns (iterations as the reference counts them, (iter_out, iter_in_sp, iter_in_dp))."""
        it, a, b, c = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _ck(self.lib.tmhip_rg_mixed_cg_her(self.h, P.h, Q.h, max_iter, eps_sq, rel_prec, N, OPS[op], delta, C.byref(it),
                                           C.byref(a), C.byref(b), C.byref(c)), "rg_mixed_cg_her")
        return it.value, (a.value, b.value, c.value)

    # --- stencil ----------------------------------------------------------
    def Hopping_Matrix(self, ieo, l, k):
        _ck(self.lib.tmhip_hopping_matrix(self.h, ieo, l.h, k.h), "Hopping_Matrix")

    def Hopping_Matrix_nocom(self, ieo, l, k):
        _ck(self.lib.tmhip_hopping_matrix_nocom(self.h, ieo, l.h, k.h), "Hopping_Matrix_nocom")

    def tm_times_Hopping_Matrix(self, ieo, l, k, c):
        _ck(self.lib.tmhip_tm_times_hopping_matrix(self.h, ieo, l.h, k.h, c.real, c.imag), "tm_times_Hopping_Matrix")

    def tm_sub_Hopping_Matrix(self, ieo, l, p, k, c):
        _ck(self.lib.tmhip_tm_sub_hopping_matrix(self.h, ieo, l.h, p.h, k.h, c.real, c.imag), "tm_sub_Hopping_Matrix")

    def D_psi(self, P, Q):
        _ck(self.lib.tmhip_D_psi(self.h, P.h, Q.h), "D_psi")

    # --- site-diagonal ----------------------------------------------------
    def mul_one_pm_imu_inv(self, l, sign, N):
        _ck(self.lib.tmhip_mul_one_pm_imu_inv(self.h, l.h, sign, N), "mul_one_pm_imu_inv")

    def assign_mul_one_pm_imu_inv(self, l, k, sign, N):
        _ck(self.lib.tmhip_assign_mul_one_pm_imu_inv(self.h, l.h, k.h, sign, N), "assign_mul_one_pm_imu_inv")

    def assign_mul_one_pm_imu(self, l, k, sign, N):
        _ck(self.lib.tmhip_assign_mul_one_pm_imu(self.h, l.h, k.h, sign, N), "assign_mul_one_pm_imu")

    def mul_one_pm_imu(self, l, sign):
        _ck(self.lib.tmhip_mul_one_pm_imu(self.h, l.h, sign), "mul_one_pm_imu")

    def mul_one_pm_imu_sub_mul(self, l, k, j, sign, N):
        _ck(self.lib.tmhip_mul_one_pm_imu_sub_mul(self.h, l.h, k.h, j.h, sign, N), "mul_one_pm_imu_sub_mul")

    def mul_one_pm_imu_sub_mul_gamma5(self, l, k, j, sign):
        _ck(self.lib.tmhip_mul_one_pm_imu_sub_mul_gamma5(self.h, l.h, k.h, j.h, sign), "mul_one_pm_imu_sub_mul_gamma5")

    def mul_one_sub_mul_gamma5(self, l, k, j):
        _ck(self.lib.tmhip_mul_one_sub_mul_gamma5(self.h, l.h, k.h, j.h), "mul_one_sub_mul_gamma5")

    def gamma5(self, l, k, N):
        _ck(self.lib.tmhip_gamma5(self.h, l.h, k.h, N), "gamma5")

    # --- compositions -----------------------------------------------------
    def H_eo_tm_inv_psi(self, l, k, ieo, sign):
        _ck(self.lib.tmhip_H_eo_tm_inv_psi(self.h, l.h, k.h, ieo, sign), "H_eo_tm_inv_psi")

    def op(self, name, l, k):
        _ck(getattr(self.lib, "tmhip_" + name)(self.h, l.h, k.h), name)

    def Qtm_pm_psi(self, l, k):
        self.op("Qtm_pm_psi", l, k)

    def M_full(self, en, on, e, o):
        _ck(self.lib.tmhip_M_full(self.h, en.h, on.h, e.h, o.h), "M_full")

    # --- linalg -----------------------------------------------------------
    def square_norm(self, P, N, parallel=0):
        out = C.c_double()
        _ck(self.lib.tmhip_square_norm(self.h, P.h, N, parallel, C.byref(out)), "square_norm")
        return out.value

    def scalar_prod_r(self, S, R, N, parallel=0):
        out = C.c_double()
        _ck(self.lib.tmhip_scalar_prod_r(self.h, S.h, R.h, N, parallel, C.byref(out)), "scalar_prod_r")
        return out.value

    def assign_add_mul_r(self, P, Q, c, N):
        _ck(self.lib.tmhip_assign_add_mul_r(self.h, P.h, Q.h, c, N), "assign_add_mul_r")

    def assign_mul_add_r(self, R, c, S, N):
        _ck(self.lib.tmhip_assign_mul_add_r(self.h, R.h, c, S.h, N), "assign_mul_add_r")

    def assign_mul_add_r_and_square(self, R, c, S, N, parallel=0):
        out = C.c_double()
        _ck(self.lib.tmhip_assign_mul_add_r_and_square(self.h, R.h, c, S.h, N, parallel, C.byref(out)),
            "assign_mul_add_r_and_square")
        return out.value

    def diff(self, Q, R, S, N):
        _ck(self.lib.tmhip_diff(self.h, Q.h, R.h, S.h, N), "diff")

    def add(self, Q, R, S, N):
        _ck(self.lib.tmhip_add(self.h, Q.h, R.h, S.h, N), "add")

    def mul_r(self, R, c, S, N):
        _ck(self.lib.tmhip_mul_r(self.h, R.h, c, S.h, N), "mul_r")

    def assign(self, R, S, N):
        _ck(self.lib.tmhip_assign(self.h, R.h, S.h, N), "assign")

    # --- solver -----------------------------------------------------------
    def cg_her(self, P, Q, max_iter, eps_sq, rel_prec, N, op="Qtm_pm_psi"):
        it = C.c_int()
        hist = np.zeros(max(max_iter, 1), dtype=np.float64)
        _ck(self.lib.tmhip_cg_her(self.h, P.h, Q.h, max_iter, eps_sq, rel_prec, N, OPS[op], C.byref(it),
                                  hist.ctypes.data_as(C.POINTER(C.c_double)), hist.size), "cg_her")
        n = it.value if it.value > 0 else max_iter
        return it.value, hist[:n]

    # --- multi-GPU --------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        _ck(self.lib.tmhip_comm_get_unique_id(buf), "tmhip_comm_get_unique_id")
        return buf.raw

    def comm_init(self, uid):
        _ck(self.lib.tmhip_comm_init(self.h, uid), "tmhip_comm_init")

    def comm_init_shm(self, job):
        """the ring over the host-staged shared-memory transport (ranks of one node, no RCCL; `job`: the same string on every rank)"""
        _ck(self.lib.tmhip_comm_init_shm(self.h, job.encode()), "tmhip_comm_init_shm")

    def comm_init_ipc(self):
        """the direct face carrier on top of the ring (collective, after comm_init / comm_init_shm): faces are stored by the producing
        waves straight into the ring neighbours' IPC-mapped receive buffers"""
        _ck(self.lib.tmhip_comm_init_ipc(self.h), "tmhip_comm_init_ipc")

    def comm_faces_direct(self):
        """(True when the faces travel as direct stores, ranks of the job on this rank's GPU)"""
        n = C.c_int()
        v = self.lib.tmhip_comm_faces_direct(self.h, C.byref(n))
        return bool(v), n.value

    def comm_sums_direct(self):
        """True when the scalar sums over the ranks travel as direct stores into every rank's block (no communicator involved)"""
        return bool(self.lib.tmhip_comm_sums_direct(self.h))

    def comm_count(self):
        """(ranks of the face communicator, ranks of the reduction communicator) as RCCL reports them; (0, 0) without one."""
        a, b = C.c_int(), C.c_int()
        _ck(self.lib.tmhip_comm_count(self.h, C.byref(a), C.byref(b)), "tmhip_comm_count")
        return a.value, b.value

    def comm_is_split(self):
        """True: reductions on their own communicator; False: they share the face communicator; None: no communicator."""
        v = self.lib.tmhip_comm_is_split(self.h)
        return None if v < 0 else bool(v)

    def set_loopback(self, on):
        _ck(self.lib.tmhip_comm_set_loopback(self.h, int(on)), "tmhip_comm_set_loopback")

    def comm_stream_delay_ms(self, ms):
        """Test hook: hold the comm stream back for `ms` milliseconds in front of the next halo exchange (a late neighbour)."""
        _ck(self.lib.tmhip_comm_stream_delay_ms(self.h, int(ms)), "tmhip_comm_stream_delay_ms")

    # --- measurement ------------------------------------------------------
    def bench_hopping(self, f0, f1, f2, iters):
        ms = C.c_double()
        _ck(self.lib.tmhip_bench_hopping(self.h, f0.h, f1.h, f2.h, iters, C.byref(ms)), "tmhip_bench_hopping")
        return ms.value

    def event_record(self, slot):
        _ck(self.lib.tmhip_event_record(self.h, slot), "tmhip_event_record")

    def event_elapsed_ms(self, a, b):
        ms = C.c_double()
        _ck(self.lib.tmhip_event_elapsed_ms(self.h, a, b, C.byref(ms)), "tmhip_event_elapsed_ms")
        return ms.value


def multi_Hopping_Matrix(lats, ieo, ls, ks):
    """Hopping_Matrix on a T-split lattice held by several contexts of THIS process (peer-copy ring)."""
    n = len(lats)
    arr = C.c_void_p * n
    _ck(lats[0].lib.tmhip_multi_hopping_matrix(n, arr(*[l.h for l in lats]), ieo, arr(*[f.h for f in ls]),
                                               arr(*[f.h for f in ks])), "tmhip_multi_hopping_matrix")


def multi_deriv_Sb(lats, ieo, ls, ks, factor):
    """deriv_Sb on a T-split lattice held by several contexts of THIS process (peer-copy ring)."""
    n = len(lats)
    arr = C.c_void_p * n
    _ck(lats[0].lib.tmhip_multi_deriv_Sb(n, arr(*[l.h for l in lats]), ieo, arr(*[f.h for f in ls]), arr(*[f.h for f in ks]),
                                         factor), "tmhip_multi_deriv_Sb")


def multi_sw_all(lats, kappa, c_sw):
    """sw_all on a T-split lattice held by several contexts of THIS process (two-sided derivative halo by peer copies)."""
    n = len(lats)
    arr = C.c_void_p * n
    _ck(lats[0].lib.tmhip_multi_sw_all(n, arr(*[l.h for l in lats]), kappa, c_sw), "tmhip_multi_sw_all")


def multi_update_gauge(lats, step):
    """update_gauge on a T-split lattice held by several contexts of THIS process: links updated, halo slabs by peer copies, stencil copies re-sorted."""
    n = len(lats)
    arr = C.c_void_p * n
    _ck(lats[0].lib.tmhip_multi_update_gauge(n, arr(*[l.h for l in lats]), step), "tmhip_multi_update_gauge")
