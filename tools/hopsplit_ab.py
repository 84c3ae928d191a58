"""Small lattices: one thread per site (64-thread blocks) vs the eight hops of a site spread over the four waves of a block
("hopsplit" 0 / 1).  us per Hopping_Matrix launch and cg_her iterations per second (live residual: 25- minus 5-iteration solves).
Usage: python tools/hopsplit_ab.py [L ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

for L in [int(a) for a in sys.argv[1:]] or [8, 12, 16, 20, 24]:
    lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
    lat.set_gauge(syn.gauge_field(1, L, L, L, L))
    src = lat.field(syn.spinor_field_eo(3, 0, L, L, L, L))
    f0, f1, f2 = lat.field(syn.spinor_field_eo(2, 0, L, L, L, L)), lat.field(), lat.field()
    x = lat.field()
    for hs in (0, 1, 0, 1):
        lat.set_option("hopsplit", hs)
        us = np.median([lat.bench_hopping(f0, f1, f2, 200) / 400 for _ in range(3)]) * 1e3

        def solve(n):
            x.zero(); lat.sync()
            t0 = time.perf_counter()
            lat.cg_her(x, src, n, 0.0, 1, lat.Vh)
            lat.sync()
            return time.perf_counter() - t0
        solve(5); solve(25)
        ts = sum(solve(5) for _ in range(10)); tl = sum(solve(25) for _ in range(10))
        print("L=%2d hopsplit=%d  Hopping_Matrix %5.1f us/launch   cg_her %6.0f it/s" % (L, hs, us, 200 / (tl - ts)), flush=True)
    lat.close()
