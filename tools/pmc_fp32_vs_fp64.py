"""Ten launches each of Hopping_Matrix (fp64) and Hopping_Matrix_32 at 32^4 for a rocprofv3 --pmc pass (tools/pmc_fp32_vs_fp64.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = T = 32
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
src = syn.spinor_field_eo(2, 1, T, L, L, L)
k, l = lat.field(src), lat.field()
k32, l32 = lat.field32(src.astype(np.float32)), lat.field32()
for _ in range(10):
    lat.Hopping_Matrix(1, l, k)
    lat.Hopping_Matrix_32(1, l32, k32)
lat.sync()
lat.close()
