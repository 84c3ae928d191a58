"""Block order A/B at 32^4 (fp64 Hopping_Matrix, per launch; also inside Qtm_pm_psi): tile order (xcd 2, default) against the brick
orders (xcd 5: 4 blocks of 8 time-slices x 2 x-ranges; xcd 6: 2 x 4) and the slab order (3); checks bitwise equality of the results."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, L, L, L, L))
f0 = lat.field(syn.spinor_field_eo(2, 0, L, L, L, L)); f1, f2 = lat.field(), lat.field()
ref = None
for x in (2, 5, 6):
    lat.set_option("xcd", x); lat.Qtm_pm_psi(f2, f0); a = f2.download()
    if ref is None:
        ref = a
    print("xcd=%d Qtm_pm_psi bitwise equal to xcd=2: %s" % (x, np.array_equal(a, ref)), flush=True)
res = {x: ([], []) for x in (2, 5, 6, 3)}
for rnd in range(5):
    for x in res:
        lat.set_option("xcd", x)
        lat.bench_hopping(f0, f1, f2, 2)
        res[x][0].append(lat.bench_hopping(f0, f1, f2, 20) / 40)
        lat.Qtm_pm_psi(f2, f0)
        lat.event_record(0)
        for _ in range(20):
            lat.Qtm_pm_psi(f2, f0)
        lat.event_record(1)
        res[x][1].append(lat.event_elapsed_ms(0, 1) / 80)
for x in res:
    h, q = (np.median(res[x][i]) * 1e3 for i in range(2))
    print("xcd=%d  hop %.1f us (%.0f GB/s alg)   Qtm_pm/4 %.1f us" % (x, h, lat.Vh * 1536 / h / 1e3, q), flush=True)
lat.close()
