"""Split-phase path (T-split rank rehearsed on one GPU, loopback): block order / tile group sweep.  Usage: variants_split.py L T"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
f0 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L))
f1, f2 = lat.field(), lat.field()
iters = 20
base = np.median([lat.bench_hopping(f0, f1, f2, iters) / (2 * iters) for _ in range(3)]) * 1e3
print("unsplit: %.1f us per launch" % base, flush=True)
lat.set_loopback(1)
grid = [(2, 0, o, f) for o in (0, 2, 3, 4) for f in (0, 1)] + [(3, 0, 3, 0), (1, 0, 3, 0)]     # f: "split_sync" (0 flags, 1 HIP events)
res = {v: [] for v in grid}
for rnd in range(3):
    for v in grid:
        lat.set_option("xcd", v[0]); lat.set_option("tgrp", v[1]); lat.set_option("occ", v[2]); lat.set_option("split_sync", v[3])
        lat.bench_hopping(f0, f1, f2, 2)
        res[v].append(lat.bench_hopping(f0, f1, f2, iters) / (2 * iters))
for us, v in sorted((float(np.median(res[v])) * 1e3, v) for v in grid):
    print("xcd=%d tgrp=%-2d occ=%d split_sync=%d  %7.1f us per launch (%.0f %% of unsplit)" % (v + (us, 100 * base / us)), flush=True)
lat.close()
