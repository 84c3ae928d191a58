"""Workload of tools/pmc_variants.sh: a few launches of the plain fp64 / fp32 stencil for every value of one option, so that
rocprofv3's per-kernel counters can be compared between the kernel variants (their template arguments differ, so do their names).
Usage: python tools/pmc_variants_run.py <option> <v0,v1,...> [L] [T]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
T = int(sys.argv[4]) if len(sys.argv) > 4 else L
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
src = syn.spinor_field_eo(2, 0, T, L, L, L)
f0, f1, f2 = lat.field(src), lat.field(), lat.field()
g0, g1, g2 = lat.field32(src.astype(np.float32)), lat.field32(), lat.field32()
lat.mixed_cg_her(f1, f0, 1, 1e-2, 1, lat.Vh)
for v in vals:
    lat.set_option(opt, v)
    for _ in range(6):
        lat.Hopping_Matrix(0, f1, f0); lat.Hopping_Matrix(1, f2, f1)
        lat.Hopping_Matrix_32(0, g1, g0); lat.Hopping_Matrix_32(1, g2, g1)
    lat.sync()
lat.close()
