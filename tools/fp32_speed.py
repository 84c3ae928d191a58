"""fp32 vs fp64: time per Qtm_pm_psi (4 stencil launches) and per Hopping_Matrix at 32^4."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = 32
lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, L, L, L, L))
src = syn.spinor_field_eo(2, 1, L, L, L, L)
k64, l64 = lat.field(src), lat.field()
k32, l32 = lat.field32(src.astype(np.float32)), lat.field32()
sw, swi = syn.clover_blocks(3, L, L, L, L, 0.01)
lat.set_clover(sw, swi)
del sw, swi
iters = 20
for rnd in range(4):
    lat.set_option("gauge_recon", 12 if rnd == 3 else 18)
    if rnd == 3:
        print("-- gauge_recon = 12 (clover launches keep the full read)", flush=True)
    for name, fn in (("Qtm_pm_psi    fp64", lambda: lat.Qtm_pm_psi(l64, k64)), ("Qtm_pm_psi_32 fp32", lambda: lat.Qtm_pm_psi_32(l32, k32)),
                     ("Qsw_pm_psi    fp64", lambda: lat.op("Qsw_pm_psi", l64, k64)), ("Qsw_pm_psi_32 fp32", lambda: lat.Qsw_pm_psi_32(l32, k32)),
                     ("Hopping_Matrix    fp64", lambda: lat.Hopping_Matrix(1, l64, k64)), ("Hopping_Matrix_32 fp32", lambda: lat.Hopping_Matrix_32(1, l32, k32))):
        fn()
        lat.event_record(0)
        for _ in range(iters):
            fn()
        lat.event_record(1)
        us = lat.event_elapsed_ms(0, 1) / iters * 1e3
        nl = 4 if name.startswith("Q") else 1
        b = 768 if "fp32" in name else 1536
        if name.startswith("Qsw"):   # + sw_inv (1152 B) on two launches, + sw (864 B) and p (192 B) on the other two
            b = b + (1152 + 864 + 192) // 2 // (2 if "fp32" in name else 1)
        print("%-24s %8.1f us/call  %6.1f us/launch  %.0f GB/s alg" % (name, us, us / nl, lat.Vh * b / (us / nl) / 1e3), flush=True)
lat.close()
