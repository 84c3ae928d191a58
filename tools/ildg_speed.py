"""ILDG record <-> resident links at L^4 (default 32): wall time of tmhip_gauge_unpack_ildg / tmhip_gauge_pack_ildg (PCIe copy of the
604 MB record included) -- run under `rocprofv3 --kernel-trace --stats` for the kernels' own time (ildg_unpack_kernel / ildg_pack_kernel:
1.2 GB of HBM traffic each at 32^4, 64-bit)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(os.environ.get("TM_L", "32"))
lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, L, L, L, L))
for prec in (64, 32):
    for rnd in range(2):
        t0 = time.perf_counter(); rec, sums = lat.gauge_pack_ildg(prec); t1 = time.perf_counter()
        s2 = lat.gauge_unpack_ildg(rec, prec); t2 = time.perf_counter()
        assert s2 == sums
        print("%d^4 prec %d: pack %.1f ms, unpack %.1f ms (record %.0f MB, checksum %08x %08x)" % (L, prec, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rec.size / 1e6, sums[0], sums[1]), flush=True)
lat.close()
