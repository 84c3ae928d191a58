"""Tile-order / occupancy sweep of the plain stencil on the production-size lattices (48^3x96 = BASELINE configs[4],
32^3x64 = configs[3] unsplit).  Usage: python tools/variants_big.py L T"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 48
T = int(sys.argv[2]) if len(sys.argv) > 2 else 96
t0 = time.time()
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
f0 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L))
f1, f2 = lat.field(), lat.field()
print("setup %.1fs  V=%d" % (time.time() - t0, lat.V), flush=True)
keys = ("xcd", "tgrp", "occ", "block")
if os.environ.get("TM_SHORT"):
    grid = [(2, 0, 3, 256), (4, 0, 3, 256), (4, 6, 3, 256), (1, 0, 3, 256), (0, 0, 3, 256), (3, 0, 3, 256)]
else:
    grid = [(2, g, o, 256) for g in (0, 2, 3, 4, 6, 8, 12) for o in (2, 3, 4)] + [(1, 0, 3, 256), (0, 0, 3, 256), (2, 4, 3, 64), (2, 6, 3, 64)]
res = {v: [] for v in grid}
iters = 5
for rnd in range(3):
    for v in grid:
        for k, val in zip(keys, v):
            lat.set_option(k, val)
        lat.bench_hopping(f0, f1, f2, 1)
        res[v].append(lat.bench_hopping(f0, f1, f2, iters) / (2 * iters))
rows = sorted((np.median(res[v]) * 1e3, v) for v in grid)
for h, v in rows:
    print("%-52s hop %8.1f us (%.0f GB/s alg)" % (dict(zip(keys, v)), h, lat.Vh * 1536 / h / 1e3), flush=True)
# fp32 stencil and the fused CG operator under the same block orders
k32, l32 = lat.field32(syn.spinor_field_eo(2, 0, T, L, L, L).astype(np.float32)), lat.field32()
lat.set_option("occ", 3); lat.set_option("block", 256); lat.set_option("tgrp", 0)
for xcd in (2, 3, 4, 0):
    lat.set_option("xcd", xcd)
    out = []
    for name, fn, nl in (("Hopping_Matrix_32", lambda: lat.Hopping_Matrix_32(1, l32, k32), 1), ("Qtm_pm_psi", lambda: lat.Qtm_pm_psi(f2, f0), 4),
                         ("Qtm_pm_psi_32", lambda: lat.Qtm_pm_psi_32(l32, k32), 4)):
        fn(); lat.sync()
        lat.event_record(0)
        for _ in range(iters):
            fn()
        lat.event_record(1)
        out.append("%s %.1f us/launch" % (name, lat.event_elapsed_ms(0, 1) / iters / nl * 1e3))
    print("xcd=%d  " % xcd + "   ".join(out), flush=True)
lat.set_option("xcd", 2)
lat.close()
