"""cg_her on small lattices (BASELINE configs[0-1]: 8^4, 16^4): iterations per second, where launch latency rather than HBM is the bound.
Usage: small_lattice_cg.py [L ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

for L in [int(a) for a in sys.argv[1:]] or [8, 16, 24]:
    lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
    lat.set_gauge(syn.gauge_field(1, L, L, L, L))
    src = lat.field(syn.spinor_field_eo(3, 0, L, L, L, L))
    f0, f1, f2 = lat.field(syn.spinor_field_eo(2, 0, L, L, L, L)), lat.field(), lat.field()
    us = np.median([lat.bench_hopping(f0, f1, f2, 200) / 400 for _ in range(3)]) * 1e3
    print("L=%d Hopping_Matrix %.1f us/launch" % (L, us), flush=True)     # (a hipGraph replay of the loop measured the same: profiles/r01_diagnostics.md)
    x = lat.field()
    for batch in (4, 16):
        lat.set_option("cg_batch", batch)
        best = 0.0
        for rep in range(3):
            x.zero()
            t0 = time.perf_counter()
            it, hist = lat.cg_her(x, src, 400, 0.0, 1, lat.Vh)     # eps 0: runs all 400 iterations
            best = max(best, len(hist) / (time.perf_counter() - t0))
        print("L=%d Hopping_Matrix %.1f us/launch; cg_her cg_batch=%d: %.0f it/s (%.1f us per iteration, %d iterations)"
              % (L, us, batch, best, 1e6 / best, len(hist)), flush=True)
    lat.close()
