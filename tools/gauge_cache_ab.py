"""Small unsplit lattices: gauge links loaded with the streaming (non-temporal) hint ("gauge_cache" 0) or without it (1: the links stay
in the Infinity Cache between calls) by the 64-thread one-thread-per-site stencil ("hopsplit" 0; the hop-split kernel of the smallest
lattices never uses the hint); -1 is the automatic choice (gauge copy <= 200 MB).  us per Hopping_Matrix launch, cg_her iterations/s.
The first measurement of this (profiles/r02_gauge_nt_small_ab.log) compared two builds of the library instead.
Usage: python tools/gauge_cache_ab.py [L ...]   (also "8x32" for an 8 x 32^3 slab)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

for arg in sys.argv[1:] or ["12", "16", "20", "24", "8x32"]:
    T, L = (int(x) for x in arg.split("x")) if "x" in arg else (int(arg), int(arg))
    lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
    lat.set_gauge(syn.gauge_field(1, T, L, L, L))
    lat.set_option("hopsplit", 0)
    src = lat.field(syn.spinor_field_eo(3, 0, T, L, L, L))
    f0, f1, f2 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L)), lat.field(), lat.field()
    x = lat.field()
    for gc in (0, 1, -1):
        lat.set_option("gauge_cache", gc)
        us = np.median([lat.bench_hopping(f0, f1, f2, 200) / 400 for _ in range(3)]) * 1e3

        def solve(n):
            x.zero(); lat.sync()
            t0 = time.perf_counter()
            lat.cg_her(x, src, n, 0.0, 1, lat.Vh)
            lat.sync()
            return time.perf_counter() - t0
        solve(5); solve(25)
        ts = sum(solve(5) for _ in range(10)); tl = sum(solve(25) for _ in range(10))
        print("%2dx%2d^3 gauge_cache=%2d  Hopping_Matrix %5.1f us/launch   cg_her %6.0f it/s" % (T, L, gc, us, 200 / (tl - ts)), flush=True)
    lat.close()
