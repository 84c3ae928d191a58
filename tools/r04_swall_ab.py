"""sw_all (owner-computes gather kernel): block order A/B (option swall_order 0 chunk / 1 slab / 2 tile, 8 = tiles of 8 x-planes) at 32^4
and, with --big, 48^3 x 96: wall time per call over 10 calls and whether the accumulated derivative is bit-identical to order 0."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice
from tmlqcd_amd import synthetic as syn
ORDERS = [int(x) for x in os.environ.get("SWALL_ORDERS", "0,1,2,8,1,2").split(",")]
kappa, mu, c_sw = 0.125, 0.01, 1.5
for (T, L) in ((32, 32), (96, 48)) if "--big" in sys.argv else ((32, 32),):
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
    g = syn.gauge_field(1, T, L, L, L)
    lat.set_gauge(g)
    lat.sw_term(g, kappa, c_sw)
    lat.sw_invert(0, mu)
    a, b = lat.field(syn.spinor_field_eo(2, 1, T, L, L, L)), lat.field(syn.spinor_field_eo(3, 0, T, L, L, L))
    lat.swpm_zero()
    lat.sw_spinor_eo(1, a, a, 0.5)
    lat.sw_spinor_eo(0, b, b, 0.5)
    lat.sw_deriv(0, mu)
    ref = None
    for order in ORDERS:
        lat.set_option("swall_order", order)
        lat.derivative_zero()
        lat.sw_all(kappa, c_sw); lat.sync()
        d = lat.derivative() if T * L ** 3 <= 32 ** 4 else None
        t0 = time.perf_counter()
        for _ in range(10):
            lat.sw_all(kappa, c_sw)
        lat.sync()
        dt = (time.perf_counter() - t0) / 10
        same = ""
        if d is not None:
            if ref is None:
                ref = d
            same = "  derivative identical to the first order's: %s" % np.array_equal(d, ref)
        print("%dx%d^3 swall_order %d: %.1f us per call (insertion pass + gather)%s" % (T, L, order, dt * 1e6, same), flush=True)
    lat.close()
