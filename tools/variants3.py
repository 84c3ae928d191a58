"""Gauge-load cache policy experiment + traffic pricing (gauge loads dropped) for the plain stencil."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = 32
lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, L, L, L, L))
f0 = lat.field(syn.spinor_field_eo(2, 0, L, L, L, L))
f1, f2 = lat.field(), lat.field()
grid = [(-1, 0, 3), (0, 0, 3), (1, 0, 3), (2, 0, 3), (3, 0, 3), (16, 0, 3), (17, 0, 3), (18, 0, 3), (19, 0, 3),
        (2, 1, 3), (2, 0, 2), (18, 0, 2), (19, 0, 2), (-1, 0, 2)]
res = {v: [] for v in grid}
iters = 10
for rnd in range(3):
    for v in grid:
        lat.set_option("gaux", v[0]); lat.set_option("gdrop", v[1]); lat.set_option("occ", v[2])
        lat.bench_hopping(f0, f1, f2, 1)
        res[v].append(lat.bench_hopping(f0, f1, f2, iters) / (2 * iters))
for v in grid:
    h = np.median(res[v]) * 1e3
    print("gaux %3d gdrop %d occ %d : %.1f us (%.0f GB/s alg)" % (v[0], v[1], v[2], h, lat.Vh * 1536 / h / 1e3), flush=True)
lat.close()
