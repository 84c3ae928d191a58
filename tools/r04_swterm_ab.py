"""sw_term: block order A/B (option swterm_order 0 / 1) at 32^4 and 48^3 x 96 (wall time over 20 calls on resident links)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice
from tmlqcd_amd import synthetic as syn
for (T, L) in ((32, 32), (96, 48)) if "--big" in sys.argv else ((32, 32),):
    lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
    lat.set_gauge(syn.gauge_field(7, T, L, L, L))
    lat.momenta_upload(np.zeros((lat.V, 4, 8)))
    lat.update_gauge(0.0)
    ref = None
    for order in [int(o) for o in os.environ.get("SWTERM_ORDERS", "0,1,0,1").split(",")]:
        lat.set_option("swterm_order", order)
        lat.sw_term(None, 0.125, 1.5); lat.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            lat.sw_term(None, 0.125, 1.5)
        lat.sync()
        dt = (time.perf_counter() - t0) / 20
        sw = lat.get_clover(True, False)[0]
        same = ""
        if sw is not None:
            if ref is None:
                ref = sw
            same = "  identical to order 0: %s" % np.array_equal(sw, ref)
        print("%dx%d^3 swterm_order %d: %.1f us per call%s" % (T, L, order, dt * 1e6, same), flush=True)
    lat.close()
