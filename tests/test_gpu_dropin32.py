"""GPU: the reference-named fp32 symbols of the drop-in on HOST spinor32 arrays (row f1's boundary) --
Hopping_Matrix_32 (operator/Hopping_Matrix_32.c:97-127), Qtm_pm_psi_32 (operator/tm_operators_32.c:94-112), the fp32 linalg and
the conversions (linalg/*_32.c) -- through ctypes against the reference's own fp32 outputs (tests/golden/ref_hs_fields_4x4.npz: the
half-spinor build, the only configuration that has the fp32 twins), and Hopping_Matrix_32_orphaned from inside an OpenMP team."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
VP = C.c_void_p
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOL32 = 2e-6


def _p(a):
    return a.ctypes.data_as(VP)


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b.astype(np.float64)).max())


@pytest.fixture(scope="module")
def host32(host_stub):
    stub, d = host_stub
    f = np.load(os.path.join(GOLD, "ref_fields_4x4.npz"))
    h = np.load(os.path.join(GOLD, "ref_hs_fields_4x4.npz"))
    s = json.load(open(os.path.join(GOLD, "ref_hs_scalars_4x4.json")))
    d.tmlqcd_hip_finalize()                                 # (a context of another test module's lattice, if any)
    g = np.ascontiguousarray(f["gauge"])
    C.memmove(stub.stub_init(4, 4, 4, 4), _p(g), g.nbytes)
    stub.stub_boundary(s["kappa"], 0.0, 0.0, 0.0, 0.0)
    stub.stub_set_mu(s["mu"])
    d.Hopping_Matrix_32.argtypes = [C.c_int, VP, VP]
    d.Hopping_Matrix_32_orphaned.argtypes = [C.c_int, VP, VP]
    d.Qtm_pm_psi_32.argtypes = [VP, VP]
    d.square_norm_32.restype = C.c_float; d.square_norm_32.argtypes = [VP, C.c_int, C.c_int]
    d.scalar_prod_r_32.restype = C.c_float; d.scalar_prod_r_32.argtypes = [VP, VP, C.c_int, C.c_int]
    d.assign_add_mul_r_32.argtypes = [VP, VP, C.c_float, C.c_int]
    d.assign_mul_add_r_32.argtypes = [VP, C.c_float, VP, C.c_int]
    d.diff_32.argtypes = [VP, VP, VP, C.c_int]
    d.assign_to_32.argtypes = [VP, VP, C.c_int]
    d.assign_to_64.argtypes = [VP, VP, C.c_int]
    yield d, f, h, s
    d.tmlqcd_hip_finalize()


def test_fp32_symbols_against_the_reference_fp32_outputs(host32):
    d, f, h, s = host32
    N = 128
    in64 = np.ascontiguousarray(f["in"])
    in32 = np.zeros((N, 4, 3, 2), dtype=np.float32)
    d.assign_to_32(_p(in32), _p(in64), N)
    assert np.array_equal(in32, h["in32"])
    a, b = np.zeros_like(in32), np.zeros_like(in32)
    d.Hopping_Matrix_32(0, _p(a), _p(in32))
    assert rel(a, h["Heo32"]) < TOL32
    d.Hopping_Matrix_32(1, _p(b), _p(a))
    assert rel(b, h["HoeHeo32"]) < TOL32
    d.Qtm_pm_psi_32(_p(b), _p(in32))
    assert rel(b, h["Qtm_pm_psi_32"]) < 2 * TOL32
    assert abs(d.square_norm_32(_p(in32), N, 1) - s["square_norm_32_in"]) <= 1e-6 * s["square_norm_32_in"]
    assert abs(d.scalar_prod_r_32(_p(in32), _p(b), N, 1) - s["scalar_prod_r_32_in_Qpm"]) <= 1e-5 * abs(s["scalar_prod_r_32_in_Qpm"])
    # the axpys and the difference: element-wise fp32 arithmetic, the same in numpy (also with aliased operands and on a prefix)
    r, q = in32.copy(), b.copy()
    d.assign_add_mul_r_32(_p(r), _p(q), 0.375, N)
    assert rel(r, in32.astype(np.float64) + 0.375 * b.astype(np.float64)) < 2e-7              # (one fp32 rounding of an fma against the exact value)
    r = in32.copy()
    d.assign_mul_add_r_32(_p(r), -1.25, _p(q), 100)
    assert rel(r[:100], -1.25 * in32[:100].astype(np.float64) + b[:100].astype(np.float64)) < 2e-7 and np.array_equal(r[100:], in32[100:])
    r = in32.copy()
    d.assign_add_mul_r_32(_p(r), _p(r), 2.0, N)                       # R += 2 R
    assert rel(r, 3.0 * in32.astype(np.float64)) < 2e-7
    dq = np.zeros_like(in32)
    d.diff_32(_p(dq), _p(in32), _p(b), N)
    assert np.array_equal(dq, in32 - b)
    back = np.zeros((N, 4, 3, 2))
    d.assign_to_64(_p(back), _p(a), N)
    assert np.array_equal(back, a.astype(np.float64))
    assert d.square_norm_32(_p(in32), 0, 1) == 0.0                     # an empty loop in the reference


def test_orphaned_operator_called_by_every_thread_of_an_openmp_team(host32):
    """operator/tm_operators_32.c:94-112: Qtm_pm_psi_32 opens `omp parallel` and every thread calls Hopping_Matrix_32_orphaned.  Here a
    team of four does that twice in a row (H_oe of H_eo): one thread issues each device call, all meet before and after it -- the
    results are the single-threaded ones."""
    d, f, h, s = host32
    so, src = os.path.join(ROOT, "tests", "host_stub", "libompcaller.so"), os.path.join(ROOT, "tests", "host_stub", "omp_caller.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fopenmp", "-fPIC", "-shared", "-o", so, src])
    omp = C.CDLL(so)
    omp.omp_team_calls_orphaned.argtypes = [VP, C.c_int, VP, VP, VP]
    in32 = np.ascontiguousarray(h["in32"])
    a, b = np.zeros_like(in32), np.zeros_like(in32)
    team = omp.omp_team_calls_orphaned(C.cast(d.Hopping_Matrix_32_orphaned, VP), 4, _p(a), _p(in32), _p(b))
    assert team == 4
    assert rel(a, h["Heo32"]) < TOL32 and rel(b, h["HoeHeo32"]) < TOL32
