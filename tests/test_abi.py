"""CPU: the C-ABI libraries load and export every symbol their headers declare (no compute calls)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tmlqcd_amd", "lib")


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//.*", "", txt)
    txt = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    txt = re.sub(r"typedef[^;]*;", "", txt)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", txt)
    return sorted(set(n for n in names if n not in ("defined", "matrix_mult")))


def exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return set(l.split()[-1] for l in out.splitlines() if l.strip())


def test_core_library_exports_header():
    names = declared_functions("tmlqcd_hip.h")
    assert len(names) > 40
    exp = exported(os.path.join(LIB, "libtmlqcd_hip.so"))
    missing = [n for n in names if n not in exp]
    assert not missing, missing


def test_dropin_library_exports_reference_symbols():
    names = declared_functions("tmlqcd_dropin.h")
    for must in ("Hopping_Matrix", "Hopping_Matrix_nocom", "tm_times_Hopping_Matrix", "tm_sub_Hopping_Matrix", "D_psi",
                 "Qtm_pm_psi", "Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi", "M_full", "Q_full",
                 "H_eo_tm_inv_psi", "mul_one_pm_imu_inv", "assign_mul_one_pm_imu_inv", "assign_mul_one_pm_imu",
                 "mul_one_pm_imu_sub_mul", "square_norm", "scalar_prod_r", "assign_add_mul_r", "assign_mul_add_r",
                 "assign_mul_add_r_and_square", "diff", "assign", "cg_her", "gamma5", "Qtm_plus_sym_psi",
                 "Qtm_minus_sym_psi", "Mtm_plus_sym_psi", "Mtm_minus_sym_psi", "Mtm_plus_sym_dagg_psi", "Qtm_pm_sym_psi",
                 "Mtm_plus_sym_psi_nocom", "Mtm_minus_sym_psi_nocom", "Qtm_plus_sym_psi_nocom", "mixed_cg_her",
                 "Qsw_pm_psi", "clover_inv", "clover_gamma5", "Qsw_minus_psi", "Qsw_plus_psi", "Msw_full",
                 "assign_mul_one_sw_pm_imu_inv", "rg_mixed_cg_her", "deriv_Sb"):
        assert must in names, must
    exp = exported(os.path.join(LIB, "libtmlqcd_dropin.so"))
    missing = [n for n in names if n not in exp]
    assert not missing, missing


def test_core_library_loads_and_reports_version():
    import tmlqcd_amd
    lib = tmlqcd_amd.load_library()
    assert b"gfx950" in lib.tmhip_version()
    assert lib.tmhip_device_count() >= 0


def test_dropin_loads_next_to_a_host_program_stub(host_stub):
    """libtmlqcd_dropin.so resolves tmLQCD's globals from the host program (here: tests/host_stub)."""
    stub, dropin = host_stub
    for g in ("T", "LX", "VOLUME", "g_update_gauge_copy", "g_mu", "ka0", "g_gauge_field"):
        C.c_int.in_dll(stub, g)
    assert dropin.Hopping_Matrix and dropin.cg_her and dropin.tmlqcd_hip_set_residency


def test_bad_geometry_is_rejected_before_touching_the_gpu():
    """Same constraints as the reference's e/o build (mpi_init.c:784-799): even extents."""
    import tmlqcd_amd
    from tmlqcd_amd.hip import TmHipError
    for dims in ((3, 4, 4, 4), (4, 4, 4, 5), (0, 4, 4, 4)):
        with pytest.raises(TmHipError):
            tmlqcd_amd.Lattice(*dims)


def test_missing_library_fails_loudly(monkeypatch):
    import tmlqcd_amd.hip as h
    monkeypatch.setattr(h, "_LIB", None)
    monkeypatch.setattr(h, "library_path", lambda: "/nonexistent/libtmlqcd_hip.so")
    with pytest.raises(h.TmHipError):
        h.load_library()


def test_reference_caller_host_program_links_against_the_drop_in():
    """Link-level check on the CPU: reference object code (cg_her, geometry, start, ...) with the hot-path objects
    left out resolves every hot-path symbol from libtmlqcd_dropin.so; the drop-in resolves tmLQCD's globals and
    update_backward_gauge from the reference objects (oracle/Makefile target libtmhostprog.so)."""
    import subprocess, sys
    so = os.path.join(ROOT, "oracle", "_ref", "libtmhostprog.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libtmhostprog.so not built (needs /root/reference)")
    code = ("import ctypes as C; h = C.CDLL(%r, mode=C.RTLD_GLOBAL); "
            "assert h.cg_her and h.geometry and h.update_backward_gauge; print('linked')" % so)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "linked" in r.stdout, r.stderr
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    for sym in ("square_norm", "scalar_prod_r", "assign_add_mul_r", "assign_mul_add_r_and_square", "diff", "assign"):
        assert (" U " + sym) in und, sym                       # really taken from the drop-in, not from reference objects


def test_complete_reference_build_with_localized_symbols_links_against_the_drop_in():
    """The other recipe of INTEGRATION.md section 2.1 (oracle/Makefile target libtmhostprog_loc.so): the complete reference
    hot-path build with the library's symbols made local.  What the library provides is undefined in the host program (so it
    comes from the library), what it does not provide is still the reference's own definition."""
    import subprocess, sys
    so = os.path.join(ROOT, "oracle", "_ref", "libtmhostprog_loc.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libtmhostprog_loc.so not built (needs /root/reference)")
    code = "import ctypes as C; h = C.CDLL(%r, mode=C.RTLD_GLOBAL); assert h.Qsw_full and h.Block_D_psi and h.sw_term; print('linked')" % so
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "linked" in r.stdout, r.stderr
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    dfd = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    for sym in ("Hopping_Matrix", "D_psi", "assign_add_mul_r", "gamma5"):       # called across object boundaries: taken from the library
        assert (" U " + sym + "\n") in und and (" T " + sym + "\n") not in dfd, sym
    for sym in ("Qsw_pm_psi", "Qtm_pm_psi", "cg_her", "square_norm"):            # the reference's definitions are no longer visible
        assert (" T " + sym + "\n") not in dfd, sym
    for sym in ("Qsw_full", "Block_D_psi", "init_sw_fields", "sw_term", "update_backward_gauge"):
        assert (" T " + sym + "\n") in dfd, sym


def test_c_host_program_links_against_the_drop_in_at_link_time(c_host_program):
    """A C main (tests/c_host/mini_benchmark.c, the shape of benchmark.c) with `-ltmlqcd_dropin -ltmlqcd_hip` on its link line:
    every reference-named symbol it calls is bound to the drop-in library, the globals to the program itself."""
    out = subprocess.run(["nm", "-D", "--undefined-only", c_host_program], capture_output=True, text=True, check=True).stdout
    und = set(l.split()[-1] for l in out.splitlines() if l.strip())
    assert {"Hopping_Matrix", "square_norm", "tmlqcd_hip_benchmark_loop", "tmlqcd_hip_finalize"} <= und
    ldd = subprocess.run(["ldd", c_host_program], capture_output=True, text=True, check=True).stdout
    assert "libtmlqcd_dropin.so" in ldd and "libtmlqcd_hip.so" in ldd and "not found" not in ldd
    defined = subprocess.run(["nm", "-D", "--defined-only", c_host_program], capture_output=True, text=True, check=True).stdout
    for g in ("g_gauge_field", "g_update_gauge_copy", "ka0", "VOLUME"):       # exported by the executable for the library to read
        assert g in defined, g


def test_register_budget_guard_of_the_build():
    """tools/check_resources.py (a prerequisite of libtmlqcd_hip.so in the Makefile): it must FAIL on a default-dispatch stencil
    instance that spills or drops below three waves per SIMD, pass on a clean one, and the table the last build committed must
    say that nothing failed."""
    import subprocess
    import sys
    import tempfile
    tool = os.path.join(ROOT, "tools", "check_resources.py")

    def remarks(name, vgpr, scratch, occ):
        head = "x.hip:1:1: remark: "
        return "\n".join([head + "Function Name: %s [-Rpass-analysis=kernel-resource-usage]" % name,
                          head + "    TotalSGPRs: 40 [-Rpass-analysis=kernel-resource-usage]",
                          head + "    VGPRs: %d [-Rpass-analysis=kernel-resource-usage]" % vgpr,
                          head + "    AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]",
                          head + "    ScratchSize [bytes/lane]: %d [-Rpass-analysis=kernel-resource-usage]" % scratch,
                          head + "    Occupancy [waves/SIMD]: %d [-Rpass-analysis=kernel-resource-usage]" % occ,
                          head + "    LDS Size [bytes/block]: 0 [-Rpass-analysis=kernel-resource-usage]"]) + "\n"
    k = "_ZN5hop6410hop_kernelILi0ELi1ELb1ELi256ELi3ELin1ELi64EEEvNS_7HopArgsE"      # hop64::hop_kernel<0, 1, true, 256, 3, -1, 64>: the split-path instance of round 3's accident
    with tempfile.TemporaryDirectory() as d:
        for tag, text, want in (("clean", remarks(k, 156, 0, 3), 0), ("spill", remarks(k, 168, 352, 3), 1), ("fat", remarks(k, 246, 0, 2), 1)):
            f = os.path.join(d, tag + ".ru.txt")
            open(f, "w").write(text)
            r = subprocess.run([sys.executable, tool, f], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            assert r.returncode == want, (tag, r.stdout, r.stderr)
    table = open(os.path.join(ROOT, "profiles", "r04_resource_usage.txt")).read()
    assert " 0 failing" in table and "hop64::hop_kernel<0, 3, true, 256, 3, -1, 64>" in table
