"""fp32 stencil: occupancy cap / two-sites-per-thread / block order sweep at L^3 x T (default 32^4)."""
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
src = syn.spinor_field_eo(2, 1, T, L, L, L).astype(np.float32)
k32, l32 = lat.field32(src), lat.field32()
grid = list(itertools.product((1, 0), (0, 2, 3, 4, 5, 6, 8), (2, 3), (18, 12)))
res = {v: [] for v in grid}
iters = 20
for rnd in range(3):
    for v in grid:
        for name, val in zip(("fp32_pairs", "occ32", "xcd", "gauge_recon"), v):
            lat.set_option(name, val)
        lat.Hopping_Matrix_32(1, l32, k32)
        lat.event_record(0)
        for _ in range(iters):
            lat.Hopping_Matrix_32(1, l32, k32)
        lat.event_record(1)
        res[v].append(lat.event_elapsed_ms(0, 1) / iters * 1e3)
for us, v in sorted((float(np.median(res[v])), v) for v in grid):
    print("pairs=%d occ=%d xcd=%d recon=%d   %7.1f us  (%.0f GB/s on %d B/site)" % (v + (us, lat.Vh * (768 if v[3] == 18 else 576) / us / 1e3, 768 if v[3] == 18 else 576)), flush=True)
lat.close()
