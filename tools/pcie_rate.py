"""PCIe-inclusive rate of one Hopping_Matrix when the boundary hands over HOST buffers (coherent drop-in mode):
upload k (AoS) -> kernel -> download l (AoS), 32^4 fp64.  Reported in DESIGN.md §1; never `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = 32
lat = Lattice(L, L, L, L)
lat.set_gauge(syn.gauge_field(1, L, L, L, L))
k = syn.spinor_field_eo(2, 1, L, L, L, L)
dk, dl = lat.field(), lat.field()
out = np.zeros_like(k)                     # the host program's own, already touched, result array
for rep in range(4):
    t0 = time.perf_counter()
    dk.upload(k)
    t1 = time.perf_counter()
    lat.Hopping_Matrix(0, dl, dk)
    lat.sync()
    t2 = time.perf_counter()
    dl.download(out=out)
    t3 = time.perf_counter()
    print("upload %.2f ms (%.1f GB/s)  kernel %.3f ms  download %.2f ms (%.1f GB/s)  total %.2f ms -> %.3f G site-updates/s"
          % (1e3 * (t1 - t0), k.nbytes / (t1 - t0) / 1e9, 1e3 * (t2 - t1), 1e3 * (t3 - t2), k.nbytes / (t3 - t2) / 1e9,
             1e3 * (t3 - t0), lat.Vh / (t3 - t0) / 1e9), flush=True)
lat.close()
