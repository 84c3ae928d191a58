"""TEST INFRASTRUCTURE ONLY (see oracle/README.md).

solver/mixed_cg_her.c:65-202 restated statement by statement in Python over the REFERENCE'S OWN object code
(oracle/_ref/libtmref_hs.so, the default half-spinor build: f32 = Qtm_pm_psi_32, the fp32 linalg, assign_to_32/64, and
f = Qtm_pm_psi, diff, add, square_norm in fp64).  The file itself cannot be compiled here -- its two tuning globals
(mixcg_innereps, mixcg_maxinnersolverit) are defined by the flex-generated input parser -- so the control flow (restart
rule :141, defect correction :153-162, return value :186) is what is restated; every floating-point operation is
executed by reference-compiled code.  Scalars are kept in the types the reference declares (:67-69: float pro, err,
alpha_cg, beta_cg, sqnrm, sqnrm2; double sqnrm_d, sourcesquarenorm).

Used by oracle/make_golden.py (fixture tests/golden/ref_mixed_*) to pin tmhip_mixed_cg_her's iteration counts and
restart points.
"""
import numpy as np

F = np.float32


def mixed_cg_her(r, iP, iQ, max_iter, eps_sq, rel_prec, innereps=5.0e-5, max_inner_it=5000, work64=(2, 3, 4), work32=(2, 3, 4, 5)):
    """r: oracle.refbind.RefLattice(hs=True) with gauge (and its fp32 copy) set.  P = g_spinor_field[iP] (output),
    Q = g_spinor_field[iQ] (source), N = VOLUME/2.  Returns (return value of mixed_cg_her, [inner iterations j per outer
    iteration], [true residual^2 after each outer iteration])."""
    lib, N = r.lib, r.V // 2
    sp, sp32 = r.sp, r.sp32
    N_outer = max_iter // max_inner_it                                  # :82
    if N_outer < 10:                                                    # :84
        N_outer = 10
    squarenorm_d = lib.square_norm(sp(iQ), N, 1)                        # :98
    sourcesquarenorm = squarenorm_d
    sqnrm_d = squarenorm_d
    delta, y, xhigh = work64                                            # :102-104
    sf32 = list(work32[:3])
    x = work32[3]                                                       # :105
    lib.assign(sp(delta), sp(iQ), N)                                    # :106
    r.spinor(iP, N)[:] = 0.0                                            # :109 zero_spinor_field
    it, js, res = 0, [], []
    for i in range(N_outer):                                            # :112
        r.spinor32(x)[:] = 0.0                                          # :115
        r.spinor32(sf32[0])[:] = 0.0                                    # :116
        lib.assign_to_32(sp32(sf32[1]), sp(delta), N)                   # :117
        lib.assign_to_32(sp32(sf32[2]), sp(delta), N)                   # :118
        sqnrm = F(sqnrm_d)                                              # :120
        sqnrm2 = sqnrm
        j = 0
        while True:                                                     # :124 for(j = 0; j <= max_inner_it; j++)
            lib.Qtm_pm_psi_32(sp32(sf32[0]), sp32(sf32[2]))             # :126
            pro = F(lib.scalar_prod_r_32(sp32(sf32[2]), sp32(sf32[0]), N, 1))
            alpha_cg = F(sqnrm2 / pro)                                  # :128 (float / float)
            lib.assign_add_mul_r_32(sp32(x), sp32(sf32[2]), alpha_cg, N)
            lib.assign_mul_add_r_32(sp32(sf32[0]), F(-alpha_cg), sp32(sf32[1]), N)
            err = F(lib.square_norm_32(sp32(sf32[0]), N, 1))            # :134
            # :141 -- mixcg_innereps is a double, sqnrm a float: the products are formed in double
            if (float(err) <= innereps * float(sqnrm)) or (j == max_inner_it) or \
               ((1.3 * float(err) <= eps_sq) and rel_prec == 0) or ((1.3 * float(err) <= eps_sq * sourcesquarenorm) and rel_prec == 1):
                break
            beta_cg = F(err / sqnrm2)                                   # :144
            lib.assign_mul_add_r_32(sp32(sf32[2]), beta_cg, sp32(sf32[0]), N)
            sf32[0], sf32[1] = sf32[1], sf32[0]                         # :146-148
            sqnrm2 = err
            j += 1
            if j > max_inner_it:                                        # loop condition of :124
                break
        it += j                                                         # :152
        js.append(j)
        lib.assign_to_64(sp(xhigh), sp32(x), N)                         # :158
        lib.add(sp(iP), sp(iP), sp(xhigh), N)                           # :159
        lib.Qtm_pm_psi(sp(y), sp(iP))                                   # :160
        lib.diff(sp(delta), sp(iQ), sp(y), N)                           # :161
        sqnrm_d = lib.square_norm(sp(delta), N, 1)                      # :162
        res.append(sqnrm_d)
        if (sqnrm_d <= eps_sq and rel_prec == 0) or (sqnrm_d <= eps_sq * sourcesquarenorm and rel_prec == 1):   # :171
            return it + i, js, res                                      # :194
        it += 1                                                         # :196
    return -1, js, res                                                  # :200
