"""Three calls of sw_all (owner-computes kernel) at 32^4 for a profiler pass (tools/pmc_kernel.sh tools/swall_once.py sw_all_gather)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(os.environ.get("TM_L", "32"))
kappa, mu, c_sw = 0.125, 0.01, 1.5
lat = Lattice(L, L, L, L, kappa=kappa, mu=mu)
g = syn.gauge_field(1, L, L, L, L)
lat.set_gauge(g)
lat.sw_term(g, kappa, c_sw)
lat.sw_invert(0, mu)
a, b = lat.field(syn.spinor_field_eo(2, 1, L, L, L, L)), lat.field(syn.spinor_field_eo(3, 0, L, L, L, L))
lat.swpm_zero()
lat.sw_spinor_eo(1, a, a, 0.5)
lat.sw_spinor_eo(0, b, b, 0.5)
lat.sw_deriv(0, mu)
lat.derivative_zero()
for _ in range(3):
    lat.sw_all(kappa, c_sw)
lat.sync()
lat.close()
