"""TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/*.npz|json from the reference's own
object code (oracle/_ref/libtmref.so, built in place from /root/reference by `make -C oracle ref`).

Each lattice size runs in its own subprocess because the reference keeps its state in globals.
Fixtures are data only: inputs (RANLUX-seeded random_gauge_field / random_spinor_field_eo,
benchmark.c:247-259, seed 123456) and the reference's outputs.

    python oracle/make_golden.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gen(T, L, full):
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle.refbind import RefLattice
    kappa, mu = 0.125, 0.01
    r = RefLattice(T, L, L, L, kappa=kappa, mu=mu, nfields=14)
    r.random_fields(123456)
    lib, N, V = r.lib, r.V // 2, r.V
    sp = r.sp
    scal = {"T": T, "L": L, "kappa": kappa, "mu": mu, "seed": 123456}
    scal["norm_in"] = lib.square_norm(sp(0), N, 0)
    lib.Hopping_Matrix(0, sp(1), sp(0))
    scal["norm_Heo"] = lib.square_norm(sp(1), N, 0)
    lib.Hopping_Matrix(1, sp(2), sp(1))
    scal["norm_HoeHeo"] = lib.square_norm(sp(2), N, 0)
    scal["HoeHeo_site0_s0c0"] = [float(x) for x in r.spinor(2, N)[0, 0, 0]]
    arrs = {}
    if full:
        arrs["gauge"] = r.gauge().copy()
        arrs["in"] = r.spinor(0, N).copy()
        arrs["Heo"] = r.spinor(1, N).copy()
        arrs["HoeHeo"] = r.spinor(2, N).copy()
        c = 0.83 - 0.41j
        lib.tm_times_Hopping_Matrix(1, sp(3), sp(1), c.real, c.imag)
        arrs["tm_times_OE_of_Heo"] = r.spinor(3, N).copy()
        lib.tm_sub_Hopping_Matrix(1, sp(3), sp(0), sp(1), c.real, c.imag)
        arrs["tm_sub_OE_p_in_k_Heo"] = r.spinor(3, N).copy()
        scal["cfactor"] = [c.real, c.imag]
        for name in ("Qtm_pm_psi", "Qtm_plus_psi", "Qtm_minus_psi", "Mtm_plus_psi", "Mtm_minus_psi"):
            getattr(lib, name)(sp(3), sp(0))
            arrs[name] = r.spinor(3, N).copy()
        # M_full on (even = in, odd = Heo)
        lib.M_full(sp(4), sp(5), sp(0), sp(1))
        arrs["M_full_even"] = r.spinor(4, N).copy()
        arrs["M_full_odd"] = r.spinor(5, N).copy()
        # D_psi on a lexicographic field built from the two parities
        e2l = r.eo2lexic()
        lex = np.zeros((V, 4, 3, 2))
        lex[e2l[:N]] = r.spinor(0, N)
        lex[e2l[N:2 * N]] = r.spinor(1, N)
        r.spinor(6, V)[:] = lex
        lib.D_psi(sp(7), sp(6))
        arrs["D_psi_in_lexic"] = lex
        arrs["D_psi_out_lexic"] = r.spinor(7, V).copy()
        arrs["eo2lexic"] = e2l.copy()
        # linalg
        scal["scalar_prod_r_in_Qpm"] = lib.scalar_prod_r(sp(0), sp(3), N, 0)
    # CG on Qtm_pm_psi, source = in, eps_sq 1e-20 relative
    lib.assign(sp(8), sp(0), N)
    r.spinor(9)[:] = 0
    it = lib.cg_her(sp(9), sp(8), 1000, 1e-20, 1, N, r.fnptr("Qtm_pm_psi"))
    scal["cg_iters"] = it
    lib.Qtm_pm_psi(sp(10), sp(9))
    lib.diff(sp(10), sp(8), sp(10), N)
    scal["cg_true_res_sq"] = lib.square_norm(sp(10), N, 0)
    scal["cg_sol_norm"] = lib.square_norm(sp(9), N, 0)
    if full:
        arrs["cg_solution"] = r.spinor(9, N).copy()
    os.makedirs(GOLD, exist_ok=True)
    tag = "%dx%d" % (T, L)
    json.dump(scal, open(os.path.join(GOLD, "ref_scalars_%s.json" % tag), "w"), indent=1)
    if full:
        np.savez_compressed(os.path.join(GOLD, "ref_fields_%s.npz" % tag), **arrs)
    print(tag, scal)


def gen_clover(T, L):
    """Clover twisted mass (invert_clover_eo path): sw_term / sw_invert outputs and the operators built on them."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle.refbind import RefLattice
    kappa, mu, c_sw = 0.125, 0.01, 1.2
    r = RefLattice(T, L, L, L, kappa=kappa, mu=mu, nfields=14)
    r.random_fields(123456)
    sw, swi = r.clover(c_sw, mu)
    lib, N = r.lib, r.V // 2
    arrs = {"sw": sw.copy(), "sw_inv": swi.copy()}
    scal = {"T": T, "L": L, "kappa": kappa, "mu": mu, "c_sw": c_sw, "seed": 123456}
    lib.Qsw_pm_psi(r.sp(1), r.sp(0)); arrs["Qsw_pm_psi"] = r.spinor(1, N).copy()
    lib.Msw_plus_psi(r.sp(1), r.sp(0)); arrs["Msw_plus_psi"] = r.spinor(1, N).copy()
    lib.assign(r.sp(2), r.sp(0), N); lib.clover_inv(r.sp(2), -1, mu); arrs["clover_inv_minus"] = r.spinor(2, N).copy()
    lib.assign(r.sp(2), r.sp(0), N); lib.clover_inv(r.sp(2), +1, mu); arrs["clover_inv_plus"] = r.spinor(2, N).copy()
    lib.Hopping_Matrix(1, r.sp(3), r.sp(0))
    lib.clover_gamma5(1, r.sp(4), r.sp(0), r.sp(3), -mu); arrs["clover_gamma5_OO_in_Hoe"] = r.spinor(4, N).copy()
    lib.assign(r.sp(8), r.sp(0), N)
    r.spinor(9)[:] = 0
    it = lib.cg_her(r.sp(9), r.sp(8), 1000, 1e-20, 1, N, r.fnptr("Qsw_pm_psi"))
    scal["cg_iters"] = it
    arrs["cg_solution"] = r.spinor(9, N).copy()
    tag = "%dx%d" % (T, L)
    json.dump(scal, open(os.path.join(GOLD, "ref_clover_scalars_%s.json" % tag), "w"), indent=1)
    np.savez_compressed(os.path.join(GOLD, "ref_clover_fields_%s.npz" % tag), **arrs)
    print("clover", tag, scal)


SYM_OPS = ("Qtm_plus_sym_psi", "Qtm_minus_sym_psi", "Mtm_plus_sym_psi", "Mtm_minus_sym_psi", "Mtm_plus_sym_dagg_psi",
           "Qtm_pm_sym_psi")


def gen_sym(T, L):
    """Symmetric e/o preconditioning family (tm_operators.c:186-364) on the same gauge / source as ref_fields_*."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle.refbind import RefLattice
    kappa, mu = 0.125, 0.01
    r = RefLattice(T, L, L, L, kappa=kappa, mu=mu, nfields=14)
    r.random_fields(123456)
    N = r.V // 2
    arrs = {}
    for name in SYM_OPS:
        getattr(r.lib, name)(r.sp(3), r.sp(0))
        arrs[name] = r.spinor(3, N).copy()
    np.savez_compressed(os.path.join(GOLD, "ref_sym_fields_%dx%d.npz" % (T, L)), **arrs)
    print("sym", T, L, {k: float((v ** 2).sum()) for k, v in arrs.items()})


def gen_force(T, L):
    """deriv_Sb.c:401 on the gauge field / source of ref_fields_*: l = in, k = Heo (as det_derivative pairs X_o with
    the hopped field), EO then OE with different factors, accumulated."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle.refbind import RefLattice
    r = RefLattice(T, L, L, L, kappa=0.125, mu=0.01, nfields=14)
    r.random_fields(123456)
    r.lib.Hopping_Matrix(0, r.sp(1), r.sp(0))
    r.deriv_Sb(1, 0, 1, 0.5)      # l on odd sites... parity labels are the caller's business: OE
    r.deriv_Sb(0, 1, 0, -0.25)    # EO with the roles swapped
    np.savez_compressed(os.path.join(GOLD, "ref_force_%dx%d.npz" % (T, L)), derivative=r.derivative().copy())
    print("force", T, L, float(np.abs(r.derivative()).max()))


def gen_rg(T, L, full):
    """solver/rg_mixed_cg_her.c:180 run by the reference's default (half-spinor) build: iteration counts for two
    values of mcg_delta, and the 4^4 solution."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle.refbind import RefLattice
    kappa, mu = 0.125, 0.01
    r = RefLattice(T, L, L, L, kappa=kappa, mu=mu, nfields=16, hs=True)
    r.random_fields(123456)
    lib, N = r.lib, r.V // 2
    lib.tmref_convert_gauge_32()
    scal = {"T": T, "L": L, "kappa": kappa, "mu": mu, "seed": 123456, "eps_sq": 1e-20, "rel_prec": 1, "runs": []}
    arrs = {}
    for delta in (0.1, 5.0e-5, 0.25, 0.5):   # 0.25 ends in the fp64 fail-safe, 0.5 exhausts N_outer (-1); 5e-5 = _default_mixcg_innereps, the value operator.c:125 puts into mcg_delta
        it = r.rg_mixed_cg_her(1, 0, delta, 2000, 1e-20, 1)
        lib.Qtm_pm_psi(r.sp(2), r.sp(1)); lib.diff(r.sp(2), r.sp(0), r.sp(2), N)
        res = lib.square_norm(r.sp(2), N, 0) / lib.square_norm(r.sp(0), N, 0)
        scal["runs"].append({"delta": delta, "iters": it, "true_rel_res_sq": res})
        if full:
            arrs["solution_delta_%g" % delta] = r.spinor(1, N).copy()
    tag = "%dx%d" % (T, L)
    json.dump(scal, open(os.path.join(GOLD, "ref_rg_scalars_%s.json" % tag), "w"), indent=1)
    if full:
        np.savez_compressed(os.path.join(GOLD, "ref_rg_fields_%s.npz" % tag), **arrs)
    print("rg", tag, scal)


def gen_mixed(T, L, full):
    """solver/mixed_cg_her.c:65-202 restated over the reference's own objects (oracle/mixed_cg_ref.py): return value, inner
    iteration count of every outer iteration and the true residual after each, for the default mixcg_innereps and a loose
    one; the 4^4 solution."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle.mixed_cg_ref import mixed_cg_her
    from oracle.refbind import RefLattice
    kappa, mu = 0.125, 0.01
    r = RefLattice(T, L, L, L, kappa=kappa, mu=mu, nfields=16, hs=True)
    r.random_fields(123456)
    lib, N = r.lib, r.V // 2
    lib.tmref_convert_gauge_32()
    scal = {"T": T, "L": L, "kappa": kappa, "mu": mu, "seed": 123456, "eps_sq": 1e-20, "rel_prec": 1, "runs": []}
    arrs = {}
    for innereps, max_inner in ((5.0e-5, 5000), (1.0e-2, 5000), (5.0e-5, 7)):   # default; frequent restarts; inner loop cut by max_inner_it
        it, js, res = mixed_cg_her(r, 1, 0, 2000, 1e-20, 1, innereps, max_inner)
        lib.Qtm_pm_psi(r.sp(2), r.sp(1)); lib.diff(r.sp(2), r.sp(0), r.sp(2), N)
        tr = lib.square_norm(r.sp(2), N, 0) / lib.square_norm(r.sp(0), N, 0)
        scal["runs"].append({"innereps": innereps, "max_inner_it": max_inner, "iters": it, "inner_iters": js, "outer_res_sq": res, "true_rel_res_sq": tr})
        if full:
            arrs["solution_innereps_%g_maxinner_%d" % (innereps, max_inner)] = r.spinor(1, N).copy()
    tag = "%dx%d" % (T, L)
    json.dump(scal, open(os.path.join(GOLD, "ref_mixed_scalars_%s.json" % tag), "w"), indent=1)
    if full:
        np.savez_compressed(os.path.join(GOLD, "ref_mixed_fields_%s.npz" % tag), **arrs)
    print("mixed", tag, scal)


def gen_md(T, L):
    """update_gauge(step, hf) (update_gauge.c:51) of the reference's default build on its RANLUX gauge field with seeded
    Gaussian momenta: the links after one step."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import numpy as np
    from oracle.refbind import RefLattice
    r = RefLattice(T, L, L, L, kappa=0.125, mu=0.01, nfields=8, hs=True)
    r.random_fields(123456)
    mom = np.random.default_rng(20260417).standard_normal((r.V, 4, 8))
    step = 0.0371
    before = r.gauge().copy()
    r.lib.tmref_update_gauge.argtypes = [C.c_double, C.c_void_p]
    r.lib.tmref_update_gauge(step, mom.ctypes.data_as(C.c_void_p))
    after = r.gauge().copy()
    assert np.abs(after - before).max() > 1e-3
    np.savez_compressed(os.path.join(GOLD, "ref_md_%dx%d.npz" % (T, L)), momenta=mom, step=np.float64(step), gauge_after=after)
    print("md %dx%d: max link change %.3e" % (T, L, np.abs(after - before).max()))


def gen_hs(T, L):
    """Default (half-spinor) build of the reference: fp64 cross-check + fp32 twins of the mixed-precision CG."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle.refbind import RefLattice
    kappa, mu = 0.125, 0.01
    r = RefLattice(T, L, L, L, kappa=kappa, mu=mu, nfields=14, hs=True)
    r.random_fields(123456)
    lib, N = r.lib, r.V // 2
    arrs, scal = {}, {"T": T, "L": L, "kappa": kappa, "mu": mu, "seed": 123456}
    lib.Hopping_Matrix(0, r.sp(1), r.sp(0)); lib.Hopping_Matrix(1, r.sp(2), r.sp(1))
    arrs["hs_Heo"], arrs["hs_HoeHeo"] = r.spinor(1, N).copy(), r.spinor(2, N).copy()
    lib.Qtm_pm_psi(r.sp(3), r.sp(0)); arrs["hs_Qtm_pm_psi"] = r.spinor(3, N).copy()
    lib.assign_to_32(r.sp32(2), r.sp(0), N)
    arrs["in32"] = r.spinor32(2).copy()
    lib.Hopping_Matrix_32(0, r.sp32(3), r.sp32(2)); arrs["Heo32"] = r.spinor32(3).copy()
    lib.Hopping_Matrix_32(1, r.sp32(4), r.sp32(3)); arrs["HoeHeo32"] = r.spinor32(4).copy()
    lib.Qtm_pm_psi_32(r.sp32(5), r.sp32(2)); arrs["Qtm_pm_psi_32"] = r.spinor32(5).copy()
    scal["square_norm_32_in"] = float(lib.square_norm_32(r.sp32(2), N, 0))
    scal["scalar_prod_r_32_in_Qpm"] = float(lib.scalar_prod_r_32(r.sp32(2), r.sp32(5), N, 0))
    tag = "%dx%d" % (T, L)
    json.dump(scal, open(os.path.join(GOLD, "ref_hs_scalars_%s.json" % tag), "w"), indent=1)
    np.savez_compressed(os.path.join(GOLD, "ref_hs_fields_%s.npz" % tag), **arrs)
    print("hs", tag, scal)


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[3] == "hs":
        gen_hs(int(sys.argv[1]), int(sys.argv[2]))
    elif len(sys.argv) == 4 and sys.argv[3] in ("rg", "rgfull"):
        gen_rg(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "rgfull")
    elif len(sys.argv) == 4 and sys.argv[3] in ("mixed", "mixedfull"):
        gen_mixed(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "mixedfull")
    elif len(sys.argv) == 4 and sys.argv[3] == "md":
        gen_md(int(sys.argv[1]), int(sys.argv[2]))
    elif len(sys.argv) == 4 and sys.argv[3] == "force":
        gen_force(int(sys.argv[1]), int(sys.argv[2]))
    elif len(sys.argv) == 4 and sys.argv[3] == "sym":
        gen_sym(int(sys.argv[1]), int(sys.argv[2]))
    elif len(sys.argv) == 4 and sys.argv[3] == "clover":
        gen_clover(int(sys.argv[1]), int(sys.argv[2]))
    elif len(sys.argv) == 4:
        gen(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "1")
    else:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "4", "4", "hs"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "4", "4", "clover"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "4", "4", "sym"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "4", "4", "force"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "4", "4", "rgfull"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "8", "8", "rg"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "4", "4", "md"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "4", "4", "mixedfull"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "8", "8", "mixed"])
        for T, L, full in ((4, 4, 1), (8, 8, 0), (6, 4, 0)):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), str(T), str(L), str(full)])
