"""sw_all at L^4 (default 32): owner-computes kernel (default) vs the reference's scatter form with fp64 atomics
("swall_atomic" 1), same inputs; prints the time per call and the largest deviation between the two derivative fields."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(os.environ.get("TM_L", "32"))
T = int(os.environ.get("TM_T", str(L)))
kappa, mu, c_sw = 0.125, 0.01, 1.5
lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
g = syn.gauge_field(1, T, L, L, L)
lat.set_gauge(g)
lat.sw_term(g, kappa, c_sw)
lat.sw_invert(0, mu)
del g
a, b = lat.field(syn.spinor_field_eo(2, 1, T, L, L, L)), lat.field(syn.spinor_field_eo(3, 0, T, L, L, L))
lat.swpm_zero()
lat.sw_spinor_eo(1, a, a, 0.5)
lat.sw_spinor_eo(0, b, b, 0.5)
lat.sw_deriv(0, mu)
res = {}
for rnd in range(2):
    for atomic, order in ((0, 0), (0, 1), (1, 0)):
        lat.set_option("swall_atomic", atomic)
        lat.set_option("swall_order", order)
        lat.derivative_zero()
        lat.sw_all(kappa, c_sw)
        res[atomic] = lat.derivative()
        lat.sync()
        lat.event_record(0)
        n = 10
        for _ in range(n):
            lat.sw_all(kappa, c_sw)
        lat.event_record(1)
        print("%dx%d^3  sw_all %-28s %9.1f us/call" % (T, L, "scatter + fp64 atomics" if atomic else "owner-computes, order %d" % order,
                                                        lat.event_elapsed_ms(0, 1) / n * 1e3), flush=True)
print("max |gather - scatter| / max |scatter| = %.2e" % (np.abs(res[0] - res[1]).max() / np.abs(res[1]).max()))
lat.close()
