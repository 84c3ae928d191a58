"""Qsw_pm_psi (BASELINE configs[4]'s operator) on one rank's share of 48^3 x 96 over 8 GPUs (12 x 48^3): unsplit vs the split forms over the
self-loopback, fp64 and fp32; us per Qsw_pm_psi (four stencils + the clover blocks)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice
from tmlqcd_amd import synthetic as syn
T, L = int(os.environ.get("PT", "12")), int(os.environ.get("PL", "48"))
kappa, mu, c_sw = 0.125, 0.01, 1.5
g = syn.gauge_field(1, T, L, L, L)
k = syn.spinor_field_eo(2, 0, T, L, L, L)
ref = None
for name, lb, opts in (("unsplit", 0, {}), ("default form (copies)", 1, {}), ("direct, automatic", 3, {}), ("direct, one kernel forced", 3, {"direct_form": 1}), ("direct, two kernels", 3, {"direct_form": 0})):
    lat = Lattice(T, L, L, L, kappa=kappa, mu=mu)
    lat.set_gauge(g)
    lat.sw_term(g, kappa, c_sw); lat.sw_invert(0, mu)
    for o, v in opts.items():
        lat.set_option(o, v)
    if lb:
        lat.set_loopback(lb)
    a, b = lat.field(k), lat.field()
    for _ in range(int(os.environ.get("PWARM", "300"))):      # (the first lattice of a process otherwise pays the GPU's clock ramp: its numbers came out 10 - 25 % high)
        lat.op("Qsw_pm_psi", b, a)
    lat.sync()
    out = b.download()
    if ref is None:
        ref = out
    dev = np.abs(out - ref).max() / np.abs(ref).max()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        lat.op("Qsw_pm_psi", b, a)
    lat.sync()
    dt = (time.perf_counter() - t0) / n
    a32, b32 = lat.field32(k.astype(np.float32)), lat.field32()
    lat.Qsw_pm_psi_32(b32, a32); lat.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        lat.Qsw_pm_psi_32(b32, a32)
    lat.sync()
    dt32 = (time.perf_counter() - t0) / n
    print("%dx%d^3 %-28s Qsw_pm_psi %.1f us (dev vs unsplit %.1e)   Qsw_pm_psi_32 %.1f us" % (T, L, name, dt * 1e6, dev, dt32 * 1e6), flush=True)
    lat.close()
