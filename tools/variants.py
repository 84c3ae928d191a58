"""A/B timing of Hopping_Matrix kernel variants on one MI355X (interleaved rounds, one process).
Usage: python tools/variants.py [L] [T] ; prints us per launch for the plain stencil (benchmark.c loop)
and for Qtm_pm_psi (2x tm_times + 2x tm_sub epilogues)."""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else L
t0 = time.time()
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
f0 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L))
f1, f2 = lat.field(), lat.field()
print("setup %.1fs  V=%d" % (time.time() - t0, lat.V), flush=True)
keys = ("block", "minw", "xcd", "occ")
grid = list(itertools.product((64, 256), (0, 4), (1, 2), (0, 2, 3)))
res = {v: ([], []) for v in grid}
iters = 10
for rnd in range(3):
    for v in grid:
        for k, val in zip(keys, v):
            lat.set_option(k, val)
        lat.bench_hopping(f0, f1, f2, 1)
        res[v][0].append(lat.bench_hopping(f0, f1, f2, iters) / (2 * iters))
        lat.Qtm_pm_psi(f2, f0)
        lat.event_record(0)
        for _ in range(iters):
            lat.Qtm_pm_psi(f2, f0)
        lat.event_record(1)
        res[v][1].append(lat.event_elapsed_ms(0, 1) / (4 * iters))
rows = []
for v in grid:
    h, q = np.median(res[v][0]) * 1e3, np.median(res[v][1]) * 1e3
    rows.append((h, q, v))
rows.sort()
for h, q, v in rows:
    print("%-44s hop %.1f us (%.0f GB/s alg)   Qtm_pm/4 %.1f us" % (dict(zip(keys, v)), h, lat.Vh * 1536 / h / 1e3, q), flush=True)
lat.close()
