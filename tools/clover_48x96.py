"""BASELINE configs[4] in full size on one MI355X: 48^3 x 96 clover twisted mass, fp64 / fp32 operator timings and the
three solvers to |r|/|b| = 1e-10.  Gauge field: synthetic SU(3) (seeded); clover blocks computed on the device
(tmhip_sw_term / tmhip_sw_invert).  Bails out early when the box does not have the host memory for the 6.1 GB gauge
field and its temporaries."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1048576.0
    return 0.0


L, T = int(os.environ.get("TM_L", "48")), int(os.environ.get("TM_T", "96"))
V = T * L ** 3
need = 3.0 * V * 4 * 144 / 2 ** 30 + 8
print("lattice %dx%d^3  V=%d  host MemAvailable %.0f GB, need ~%.0f GB" % (T, L, V, mem_available_gb(), need), flush=True)
if mem_available_gb() < need:
    print("not enough host memory on this box; skipping", flush=True)
    sys.exit(0)
kappa, mu, c_sw = 0.1394265, 0.00072 * 2 * 0.1394265, 1.69     # cA2.09.48-like bare parameters (2 kappa mu)
t0 = time.perf_counter()
g = syn.gauge_field(1, T, L, L, L)
print("host gauge field %.1f GB generated in %.0f s" % (g.nbytes / 2 ** 30, time.perf_counter() - t0), flush=True)
lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
t0 = time.perf_counter(); lat.set_gauge(g); lat.sync(); t1 = time.perf_counter()
lat.sw_term(g, kappa, c_sw); lat.sync(); t2 = time.perf_counter()
lat.sw_invert(0, mu); lat.sync(); t3 = time.perf_counter()
print("set_gauge %.0f ms   sw_term %.0f ms (both incl. the 6.1 GB host->device copy)   sw_invert %.1f ms" %
      ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
del g
Vh = lat.Vh
src = syn.spinor_field_eo(2, 1, T, L, L, L)
a, c = lat.field(src), lat.field()
a32, c32 = lat.field32(src.astype(np.float32)), lat.field32()


def timed(name, fn, iters, nl, bps):
    fn(); lat.sync()
    lat.event_record(0)
    for _ in range(iters):
        fn()
    lat.event_record(1)
    us = lat.event_elapsed_ms(0, 1) / iters * 1e3
    print("%-22s %9.1f us/call %8.1f us/launch  %6.0f GB/s alg" % (name, us, us / nl, Vh * bps / (us / nl) / 1e3), flush=True)


timed("Hopping_Matrix fp64", lambda: lat.Hopping_Matrix(1, c, a), 10, 1, 1536)
timed("Qtm_pm_psi fp64", lambda: lat.Qtm_pm_psi(c, a), 10, 4, 1536 + 96)
timed("Qsw_pm_psi fp64", lambda: lat.op("Qsw_pm_psi", c, a), 10, 4, 1536 + (1152 + 864 + 192) // 2)
timed("Qsw_pm_psi_32 fp32", lambda: lat.Qsw_pm_psi_32(c32, a32), 10, 4, 768 + (1152 + 864 + 192) // 4)
x = lat.field()
for name, fn in (("cg_her", lambda: (x.zero(), lat.cg_her(x, a, 20000, 1e-20, 1, Vh, op="Qsw_pm_psi")[0])[1]),
                 ("mixed_cg_her", lambda: lat.mixed_cg_her(x, a, 20000, 1e-20, 1, Vh, op="Qsw_pm_psi")),
                 ("rg_mixed_cg_her 5e-5", lambda: lat.rg_mixed_cg_her(x, a, 20000, 1e-20, 1, Vh, delta=5e-5, op="Qsw_pm_psi"))):
    lat.sync()
    t0 = time.perf_counter(); r = fn(); lat.sync(); t1 = time.perf_counter()
    lat.op("Qsw_pm_psi", c, x); lat.diff(c, a, c, Vh)
    res = lat.square_norm(c, Vh) / lat.square_norm(a, Vh)
    print("%-22s -> %-26s %9.1f ms   true |r|^2/|b|^2 = %.2e" % (name, r, (t1 - t0) * 1e3, res), flush=True)
lat.close()
