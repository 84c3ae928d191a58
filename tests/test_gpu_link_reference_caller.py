"""GPU, link-level integration: the REFERENCE's own caller code drives the drop-in library.

oracle/_ref/libtmhostprog.so is a stand-in executable made of reference objects only (solver/cg_her.c,
solver/solver_field.c, geometry_eo.c, boundary.c, start.c + RANLUX, init_gauge_field.c,
update_backward_gauge.c, globals) with every hot-path object left out and libtmlqcd_dropin.so on its link
line.  So the reference's cg_her -- unmodified object code -- calls OUR square_norm,
diff, assign, scalar_prod_r, assign_add_mul_r, assign_mul_add_r_and_square, assign_mul_add_r and, through
the matrix_mult pointer, OUR Qtm_pm_psi, all with host AoS pointers in the default coherent mode, and our
Hopping_Matrix calls THEIR update_backward_gauge when g_update_gauge_copy is set."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTPROG = os.path.join(ROOT, "oracle", "_ref", "libtmhostprog.so")

CHILD = r'''
import ctypes as C, json, sys
sys.path.insert(0, %(root)r)
import numpy as np
host = C.CDLL(%(hostprog)r, mode=C.RTLD_GLOBAL)          # pulls in libtmlqcd_dropin.so + libtmlqcd_hip.so
dropin = C.CDLL(%(root)r + "/tmlqcd_amd/lib/libtmlqcd_dropin.so", mode=C.RTLD_GLOBAL)
host.tmref_init.argtypes = [C.c_int] * 4 + [C.c_double] * 2 + [C.c_int] * 2
host.tmref_spinor.restype = C.c_void_p; host.tmref_spinor.argtypes = [C.c_int]
host.tmref_gauge.restype = C.c_void_p
L = 8
assert host.tmref_init(L, L, L, L, 0.125, 0.01, 14, 1) == 0
host.tmref_random_fields(123456)                          # benchmark.c:247-259
V, N = L ** 4, L ** 4 // 2
sp = host.tmref_spinor
def view(i, n=N):
    return np.frombuffer((C.c_double * (n * 24)).from_address(sp(i)), dtype=np.float64).reshape(n, 4, 3, 2)
gauge = np.frombuffer((C.c_double * (V * 72)).from_address(host.tmref_gauge()), dtype=np.float64).reshape(V, 4, 3, 3, 2).copy()
src = view(0).copy()
view(1)[:] = 0
host.cg_her.restype = C.c_int
host.cg_her.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
f = C.cast(dropin.Qtm_pm_psi, C.c_void_p)
assert C.c_int.in_dll(host, "g_update_gauge_copy").value == 1
it = host.cg_her(sp(1), sp(0), 1000, 1e-20, 1, N, f)     # the reference's cg_her object code
assert C.c_int.in_dll(host, "g_update_gauge_copy").value == 0   # their update_backward_gauge ran and cleared it
sol = view(1).copy()
# independent fp64 check on the CPU oracle
from oracle.oraclebind import Oracle
o = Oracle(L, L, L, L, kappa=0.125, mu=0.01)
o.set_gauge(gauge)
full = o.new_field(); full[:N] = sol
chk = o.new_field(); o.op("Qtm_pm_psi", chk, full)
res = float(((chk[:N] - src) ** 2).sum() / (src ** 2).sum())
Pref = o.new_field(); it_ref, _ = o.cg_her(Pref, src.copy(), 1000, 1e-20, 1, N)
err = float(np.abs(sol - Pref[:N]).max() / np.abs(Pref[:N]).max())
# reference-compiled caller builds solver_params_t (junk except mcg_delta) and passes it BY VALUE to our
# rg_mixed_cg_her (solver/rg_mixed_cg_her.c:180); f32 travels on the stack behind the 144-byte struct
host.tmref_rg_mixed_cg_her.restype = C.c_int
host.tmref_rg_mixed_cg_her.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int]
view(2)[:] = 7.0                                           # the solver must zero its result field itself
it_rg = host.tmref_rg_mixed_cg_her(sp(2), sp(0), 0.1, 1000, 1e-20, 1, N, 0)
err_rg = float(np.abs(view(2) - Pref[:N]).max() / np.abs(Pref[:N]).max())
dropin.tmlqcd_hip_finalize()
print(json.dumps({"iters": it, "iters_oracle": it_ref, "true_res_rel": res, "sol_err": err, "iters_rg": it_rg, "sol_err_rg": err_rg}))
'''


@pytest.mark.skipif(not os.path.exists(HOSTPROG), reason="oracle/_ref/libtmhostprog.so not built (needs /root/reference)")
def test_reference_cg_her_object_code_drives_the_drop_in():
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "hostprog": HOSTPROG}],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_scalars_8x8.json")))
    assert abs(d["iters"] - gold["cg_iters"]) <= 1           # 36 iterations in the all-reference run
    assert abs(d["iters"] - d["iters_oracle"]) <= 1
    assert d["true_res_rel"] <= 4e-20 and d["sol_err"] < 1e-8
    rg = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_rg_scalars_8x8.json")))["runs"][0]
    assert rg["delta"] == 0.1 and abs(d["iters_rg"] - rg["iters"]) <= 6      # 59 in the all-reference run
    assert d["sol_err_rg"] < 1e-8


def test_c_main_linked_at_link_time_runs_the_benchmark_loop(c_host_program):
    """tests/c_host/mini_benchmark (C main + tmLQCD globals, `-ltmlqcd_dropin -ltmlqcd_hip` at link time): benchmark.c's loop on host
    arrays in coherent mode and resident in HBM, checked inside the program against the oracle."""
    r = subprocess.run([c_host_program, "8", "8", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["max_rel_err_vs_oracle"] < 1e-13 and d["mflops_resident"] > d["mflops_coherent"] > 0


def test_unmodified_c_main_runs_resident_under_the_lazy_mode(c_host_program):
    """The same binary, not a byte changed, with TMLQCD_HIP_RESIDENCY=lazy in its environment: its plain loop over the reference's
    symbols (host calloc'ed arrays, no residency call) keeps the fields in HBM -- the program's own loads decide when a page comes
    back -- and still passes its built-in comparison with the oracle (it reads the complete output on the host afterwards)."""
    exe = c_host_program
    out = {}
    for mode in ("coherent", "lazy"):
        env = dict(os.environ, TMLQCD_HIP_RESIDENCY=mode)
        # (the lazy run pays once for its page-locked bounce buffer and the first upload: enough iterations to see the steady state)
        r = subprocess.run([exe, "32", "32", "400" if mode == "lazy" else "20"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
        assert out[mode]["max_rel_err_vs_oracle"] < 1e-13
    print("mini_benchmark 32^4, plain loop over host arrays: coherent %.0f Mflop/s, lazy %.0f Mflop/s, resident helper %.0f Mflop/s"
          % (out["coherent"]["mflops_coherent"], out["lazy"]["mflops_coherent"], out["lazy"]["mflops_resident"]))
    # (a sanity bound, not a benchmark: 35x and 0.7 measured; the lazy run's one-time costs -- page-locked bounce buffer, first
    # upload, first protection of three 100 MB arrays -- vary with the state of the box)
    assert out["lazy"]["mflops_coherent"] > 3 * out["coherent"]["mflops_coherent"]
    assert out["lazy"]["mflops_coherent"] > 0.25 * out["lazy"]["mflops_resident"]


HOSTPROG_LOC = os.path.join(ROOT, "oracle", "_ref", "libtmhostprog_loc.so")

CHILD_LOC = r'''
import ctypes as C, json, sys
sys.path.insert(0, %(root)r)
import numpy as np
host = C.CDLL(%(hostprog)r, mode=C.RTLD_GLOBAL)
dropin = C.CDLL(%(root)r + "/tmlqcd_amd/lib/libtmlqcd_dropin.so", mode=C.RTLD_GLOBAL)
dropin.tmlqcd_hip_calls.restype = C.c_ulong
host.tmref_init.argtypes = [C.c_int] * 4 + [C.c_double] * 2 + [C.c_int] * 2
host.tmref_spinor.restype = C.c_void_p; host.tmref_spinor.argtypes = [C.c_int]
host.tmref_gauge.restype = C.c_void_p
host.tmref_clover.argtypes = [C.c_double, C.c_double]
L, kappa, mu, c_sw = 8, 0.125, 0.02, 1.3
assert host.tmref_init(L, L, L, L, kappa, mu, 14, 1) == 0
host.tmref_random_fields(4711)
host.tmref_clover(c_sw, mu)                               # the reference's sw_term / sw_invert fill ITS sw / sw_inv arrays
V, N = L ** 4, L ** 4 // 2
sp = host.tmref_spinor
def view(i, n=N):
    return np.frombuffer((C.c_double * (n * 24)).from_address(sp(i)), dtype=np.float64).reshape(n, 4, 3, 2)
gauge = np.frombuffer((C.c_double * (V * 72)).from_address(host.tmref_gauge()), dtype=np.float64).reshape(V, 4, 3, 3, 2).copy()
rng = np.random.default_rng(5)
view(0)[:] = rng.standard_normal((N, 4, 3, 2)); view(1)[:] = rng.standard_normal((N, 4, 3, 2))
e, o_ = view(0).copy(), view(1).copy()
# Qsw_full is NOT in the drop-in library: the reference's object code runs, and what it calls across object boundaries
# (Hopping_Matrix, assign_add_mul_r, gamma5) must come from the library, what it calls inside its own object stays CPU code
assert not hasattr(dropin, "Qsw_full_does_not_exist")
n0 = dropin.tmlqcd_hip_calls()
host.Qsw_full.argtypes = [C.c_void_p] * 4
host.Qsw_full(sp(2), sp(3), sp(0), sp(1))
served = dropin.tmlqcd_hip_calls() - n0
en, on = view(2).copy(), view(3).copy()
from oracle.oraclebind import Oracle
orc = Oracle(L, L, L, L, kappa=kappa, mu=mu)
orc.set_gauge(gauge)
sw = orc.sw_term(kappa, c_sw); swi, _ = orc.sw_invert(sw, 0, mu)
orc.set_clover(sw, swi); orc.set_mu(mu)
fe, fo, re_, ro = orc.new_field(), orc.new_field(), orc.new_field(), orc.new_field()
fe[:N] = e; fo[:N] = o_
orc.Msw_full(re_, ro, fe, fo)
g5 = np.array([1.0, 1.0, -1.0, -1.0]).reshape(1, 4, 1, 1)
err = max(float(np.abs(en - g5 * re_[:N]).max()), float(np.abs(on - g5 * ro[:N]).max())) / float(np.abs(re_[:N]).max())
dropin.tmlqcd_hip_finalize()
print(json.dumps({"served": int(served), "err": err}))
'''


@pytest.mark.skipif(not os.path.exists(HOSTPROG_LOC), reason="oracle/_ref/libtmhostprog_loc.so not built (needs /root/reference)")
def test_complete_reference_build_with_replaced_symbols_localized():
    """The other recipe of INTEGRATION.md: nothing is removed from the reference build; the symbols this library provides are
    made local in the reference objects.  The reference's Qsw_full (operator/clovertm_operators.c:98, not provided here) then
    runs as object code, its two Hopping_Matrix, two assign_add_mul_r and two gamma5 calls are served by the library, its
    assign_mul_one_sw_pm_imu calls stay inside its own object, and the result is the all-CPU one."""
    r = subprocess.run([sys.executable, "-c", CHILD_LOC % {"root": ROOT, "hostprog": HOSTPROG_LOC}],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["served"] == 6, d
    assert d["err"] < 1e-13, d
