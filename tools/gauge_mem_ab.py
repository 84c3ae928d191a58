"""One-off A/B: the link stream in default / uncached / fine-grained / contiguous device memory ("gauge_mem" 0..3), with and
without the non-temporal hint ("nt").  us per Hopping_Matrix launch at 32^4 (or L T given)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else L
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
g = syn.gauge_field(1, T, L, L, L)
f0 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L))
f1, f2 = lat.field(), lat.field()
iters = 20
for rnd in range(2):
    for mem in (0, 1, 2, 3):
        try:
            lat.set_option("gauge_mem", mem)
            lat.set_gauge(g)
        except Exception as e:
            print("gauge_mem", mem, "failed:", e, flush=True)
            continue
        for nt in (1, 0):
            lat.set_option("nt", nt)
            lat.bench_hopping(f0, f1, f2, 3)
            ts = [lat.bench_hopping(f0, f1, f2, iters) / (2 * iters) * 1e3 for _ in range(3)]
            print("gauge_mem %d nt %d: %.1f us per launch" % (mem, nt, float(np.median(ts))), flush=True)
lat.close()
