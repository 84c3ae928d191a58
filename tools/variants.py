"""A/B timing of Hopping_Matrix kernel variants on one MI355X (interleaved rounds, one process).
Usage: python tools/variants.py [L] [T] ; prints ms per Hopping_Matrix call and derived rates."""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import random_gauge, random_spinor  # noqa: E402
from tmlqcd_amd import Lattice  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else L
t0 = time.time()
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
g = random_gauge(1, lat.VPR)
lat.set_gauge(g)
del g
f0 = lat.field(random_spinor(2, lat.Vh))
f1, f2 = lat.field(), lat.field()
print("setup %.1fs  V=%d" % (time.time() - t0, lat.V), flush=True)
variants = [dict(block=b, nt=n, xcd=x) for b, n, x in itertools.product((64, 128, 256), (0, 1), (0, 1))]
res = {i: [] for i in range(len(variants))}
iters = 20
for rnd in range(4):
    for i, v in enumerate(variants):
        for k, val in v.items():
            lat.set_option(k, val)
        lat.bench_hopping(f0, f1, f2, 2)
        ms = lat.bench_hopping(f0, f1, f2, iters)
        res[i].append(ms / (2 * iters))
for i, v in enumerate(variants):
    a = np.array(res[i])
    ms = np.median(a)
    sites = lat.Vh / (ms * 1e-3)
    print("%-32s median %.4f ms  min %.4f  | %.2f Gsites/s  %.0f GB/s(1536B)  %.2f Tflop/s" %
          (v, ms, a.min(), sites / 1e9, sites * 1536 / 1e9, sites * 1608 / 1e12), flush=True)
lat.close()
