"""Timings at 32^4 of the SURVEY §8f rows built after the core path: device sw_term / sw_invert, deriv_Sb,
the symmetric operator family and the three CG flavours (cg_her, mixed_cg_her, rg_mixed_cg_her) on Qtm_pm_psi."""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L = int(os.environ.get("TM_L", "32"))
kappa, mu, c_sw = 0.125, 0.01, 1.5
lat = Lattice(L, L, L, L, kappa=kappa, mu=mu)
g = syn.gauge_field(1, L, L, L, L)
lat.set_gauge(g)
Vh = lat.Vh


def timed(name, fn, iters, bytes_per_site=None, sites=None):
    fn()
    lat.sync()
    lat.event_record(0)
    for _ in range(iters):
        fn()
    lat.event_record(1)
    us = lat.event_elapsed_ms(0, 1) / iters * 1e3
    extra = ""
    if bytes_per_site:
        extra = "  %.0f GB/s alg" % ((sites or Vh) * bytes_per_site / us / 1e3)
    print("%-34s %10.1f us/call%s" % (name, us, extra), flush=True)
    return us


# clover term / inverse: wall time incl. the 604 MB host->device gauge copy of sw_term (it takes the host field)
for rnd in range(2):
    t0 = time.perf_counter(); lat.sw_term(g, kappa, c_sw); lat.sync(); t1 = time.perf_counter()
    lat.sw_invert(0, mu); lat.sync(); t2 = time.perf_counter()
    print("sw_term (incl. H2D of the gauge field) %8.1f ms   sw_invert(EE, mu) %8.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
# ... and from the links already resident in HBM (what an MD step costs after tmhip_update_gauge): kernels only
for rnd in range(2):
    t0 = time.perf_counter(); lat.sw_term(None, kappa, c_sw); lat.sync(); t1 = time.perf_counter()
    print("sw_term from the resident links %8.2f ms" % ((t1 - t0) * 1e3), flush=True)
lat.sw_invert(0, mu)
del g

src = syn.spinor_field_eo(2, 1, L, L, L, L)
a, b, c = lat.field(src), lat.field(), lat.field()
lat.Hopping_Matrix(0, b, a)
lat.derivative_zero()
# deriv_Sb: per site (either parity) own spinor 192 + 4 neighbour spinors 768 + 4 links 576 + 4 su3adj rd+wr 512 = 2048 B
timed("deriv_Sb", lambda: lat.deriv_Sb(1, a, b, 0.5), 20, 2048, 2 * Vh)
# clover part of the force: outer products (2 x 192 B spinors + 8 su3 RMW), tr-log term, and the leaf kernel with atomics
lat.swpm_zero()
timed("sw_spinor_eo", lambda: lat.sw_spinor_eo(1, a, b, 0.5), 20, 384 + 2 * 1152, Vh)
timed("sw_deriv(EE, mu)", lambda: lat.sw_deriv(0, mu), 20, 2 * 1152 + 2 * 1152, Vh)
timed("sw_all (owner-computes)", lambda: lat.sw_all(kappa, c_sw), 5)
lat.momenta_upload(__import__("numpy").zeros((lat.V, 4, 8)))
timed("update_gauge (exp(step P) U, halo, re-sort)", lambda: lat.update_gauge(0.01), 5)
timed("update_momenta", lambda: lat.update_momenta(0.01), 5)
lat.sw_term(None, kappa, c_sw); lat.sw_invert(0, mu)     # the links moved: clover blocks of the new links for the operators below
for name in ("Qtm_pm_psi", "Mtm_plus_psi", "Mtm_plus_sym_psi", "Qtm_plus_sym_psi", "Mtm_plus_sym_dagg_psi", "Qsw_pm_psi"):
    timed(name, lambda n=name: lat.op(n, c, a), 20)

x = lat.field()
def cg64():
    x.zero()                      # cg_her takes P as the initial guess; the mixed solvers zero it themselves
    return lat.cg_her(x, a, 5000, 1e-20, 1, Vh)[0]


for name, fn in (("cg_her", cg64),
                 ("mixed_cg_her", lambda: lat.mixed_cg_her(x, a, 5000, 1e-20, 1, Vh)),
                 ("rg_mixed_cg_her delta=5e-5", lambda: lat.rg_mixed_cg_her(x, a, 5000, 1e-20, 1, Vh, delta=5e-5)),
                 ("rg_mixed_cg_her delta=0.1", lambda: lat.rg_mixed_cg_her(x, a, 5000, 1e-20, 1, Vh, delta=0.1))):
    fn()
    lat.sync()
    t0 = time.perf_counter(); r = fn(); lat.sync(); t1 = time.perf_counter()
    lat.Qtm_pm_psi(c, x); lat.diff(c, a, c, Vh)
    res = lat.square_norm(c, Vh) / lat.square_norm(a, Vh)
    print("%-30s -> %-22s %8.2f ms   true |r|^2/|b|^2 = %.2e" % (name, r, (t1 - t0) * 1e3, res), flush=True)
lat.close()
