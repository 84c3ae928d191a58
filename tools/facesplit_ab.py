"""Split-phase path (loopback): face kernel A/B -- "facesplit" 1 (hops of a site over the 4 waves of a block) vs 0 (thread per site),
crossed with "fusedface".  Stencil and cg_her.  Usage: facesplit_ab.py L T [T ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for T in [int(t) for t in sys.argv[2:]] or [8]:
    lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
    lat.set_gauge(syn.gauge_field(1, T, L, L, L))
    f0 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L))
    f1, f2 = lat.field(), lat.field()
    iters = 20
    base = np.median([lat.bench_hopping(f0, f1, f2, iters) / (2 * iters) for _ in range(3)]) * 1e3
    print("T=%d unsplit: %.1f us per launch" % (T, base), flush=True)
    lat.set_loopback(1)
    grid = [(fs, ff) for fs in (0, 1) for ff in (0, 1)]
    res = {v: [] for v in grid}
    for rnd in range(3):
        for v in grid:
            lat.set_option("facesplit", v[0]); lat.set_option("fusedface", v[1])
            lat.bench_hopping(f0, f1, f2, 2)
            res[v].append(lat.bench_hopping(f0, f1, f2, iters) / (2 * iters))
    for v in grid:
        us = float(np.median(res[v])) * 1e3
        print("T=%d facesplit=%d fusedface=%d  %7.1f us per launch (%.0f %% of unsplit)" % ((T,) + v + (us, 100 * base / us)), flush=True)
    # cg_her on the split path (fused iteration): iterations per second at fixed iteration count
    import time
    src = lat.field(syn.spinor_field_eo(3, 0, T, L, L, L))
    for v in grid:
        lat.set_option("facesplit", v[0]); lat.set_option("fusedface", v[1])
        best = 0.0
        for rep in range(3):
            x = lat.field()
            t0 = time.perf_counter()
            it, hist = lat.cg_her(x, src, 300, 1e-30, 0, lat.Vh)
            best = max(best, len(hist) / (time.perf_counter() - t0))   # incl. set-up and the final true-residual check
        print("T=%d facesplit=%d fusedface=%d  cg_her %.0f it/s" % ((T,) + v + (best,)), flush=True)
    lat.close()
