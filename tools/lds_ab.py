"""A/B of the LDS-staged stencil ("lds" 1: the block's own input spinors staged once, +-y / +-z neighbours read from LDS) against
the plain gather kernel, interleaved in one process on one MI355X.  Prints us per launch for the benchmark.c loop, per launch
inside Qtm_pm_psi, the fp32 stencil, and checks that both kernels give bitwise identical fields.
Usage: python tools/lds_ab.py [L] [T]"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else L
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
src = syn.spinor_field_eo(2, 0, T, L, L, L)
f0 = lat.field(src)
f1, f2 = lat.field(), lat.field()
lat.set_option("lds", 0); lat.Qtm_pm_psi(f2, f0); a = f2.download()
for m in (1,):
    lat.set_option("lds", m); lat.Qtm_pm_psi(f2, f0); b = f2.download()
    print("Qtm_pm_psi lds=%d vs lds=0: bitwise equal %s, max abs diff %.3e" % (m, np.array_equal(a, b), np.abs(a - b).max()), flush=True)
g0 = lat.field32(src.astype(np.float32)); g1, g2 = lat.field32(), lat.field32()
lat.mixed_cg_her(f1, f0, 2, 1e-20, 1, lat.Vh)   # builds the fp32 gauge copy
keys = ("lds", "occ", "xcd", "lds32")
grid = [(0, 3, 2, 0), (1, 3, 2, 0), (1, 2, 2, 0), (1, 0, 2, 0), (1, 3, 3, 0), (0, 3, 3, 0), (1, 3, 2, 1), (0, 3, 2, 1)]
res = {v: ([], [], []) for v in grid}
iters = 20
for rnd in range(4):
    for v in grid:
        for k, val in zip(keys, v):
            lat.set_option(k, val)
        lat.bench_hopping(f0, f1, f2, 2)
        res[v][0].append(lat.bench_hopping(f0, f1, f2, iters) / (2 * iters))
        lat.Qtm_pm_psi(f2, f0)
        lat.event_record(0)
        for _ in range(iters):
            lat.Qtm_pm_psi(f2, f0)
        lat.event_record(1)
        res[v][1].append(lat.event_elapsed_ms(0, 1) / (4 * iters))
        lat.Hopping_Matrix_32(0, g1, g0)
        lat.event_record(2)
        for _ in range(iters):
            lat.Hopping_Matrix_32(0, g1, g0); lat.Hopping_Matrix_32(1, g2, g1)
        lat.event_record(3)
        res[v][2].append(lat.event_elapsed_ms(2, 3) / (2 * iters))
for v in grid:
    h, q, s = (np.median(res[v][i]) * 1e3 for i in range(3))
    print("%-48s hop %.1f us (%.0f GB/s alg)   Qtm_pm/4 %.1f us   hop32 %.1f us (%.0f GB/s alg)"
          % (dict(zip(keys, v)), h, lat.Vh * 1536 / h / 1e3, q, s, lat.Vh * 768 / s / 1e3), flush=True)
lat.close()
