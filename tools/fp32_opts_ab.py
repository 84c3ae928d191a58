"""A/B of the launch options of the fp32 stencil where it matters: inside Qtm_pm_psi_32 (the fused epilogues of the mixed CG's inner loop), 32^4."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, L, L, L, L))
src = syn.spinor_field_eo(2, 1, L, L, L, L)
k32, l32 = lat.field32(src.astype(np.float32)), lat.field32()
k64, p64 = lat.field(src), lat.field()
iters = 30


def t_q32():
    lat.Qtm_pm_psi_32(l32, k32)
    lat.event_record(0)
    for _ in range(iters):
        lat.Qtm_pm_psi_32(l32, k32)
    lat.event_record(1)
    return lat.event_elapsed_ms(0, 1) / iters * 1e3 / 4


def t_mixed():
    best = 1e9
    for _ in range(3):
        p64.zero(); lat.sync()
        t0 = time.perf_counter()
        it = lat.mixed_cg_her(p64, k64, 1000, 1e-20, 1, lat.Vh)
        lat.sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, it


for occ32 in (0, 2, 3, 4):
    for lds32 in (0, 1):
        for xcd in (2, 3):
            lat.set_option("occ32", occ32); lat.set_option("lds32", lds32); lat.set_option("xcd", xcd)
            us = t_q32()
            ms, it = t_mixed()
            print("occ32 %d lds32 %d xcd %d: Qtm_pm_psi_32 %.1f us per launch, mixed_cg_her to 1e-10: %.2f ms (%s iterations)" % (occ32, lds32, xcd, us, ms, it), flush=True)
lat.close()
