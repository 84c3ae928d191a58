"""The clover rows on a T-split rank: sw_term and sw_all of one rank of a 2-way split (two contexts of this process, T_local = PT each) against
the unsplit lattice of the same local size.  The t = 0 / T-1 slices of a split rank go through the edge kernels (raw links + halo slabs)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice
from tmlqcd_amd import synthetic as syn
from tmlqcd_amd.hip import multi_sw_all
T, L = int(os.environ.get("PT", "8")), int(os.environ.get("PL", "32"))
kappa, mu, c_sw = 0.125, 0.01, 1.5


def prep(lat, g, world, r):
    lat.set_gauge(g)
    lat.sw_term(g, kappa, c_sw); lat.sw_invert(0, mu)
    a, b = lat.field(syn.spinor_field_eo(2, 1, T, L, L, L, world, r)), lat.field(syn.spinor_field_eo(3, 0, T, L, L, L, world, r))
    lat.swpm_zero(); lat.sw_spinor_eo(1, a, a, 0.5); lat.sw_spinor_eo(0, b, b, 0.5); lat.sw_deriv(0, mu); lat.derivative_zero()


def timed(f, sync, n=20):
    for _ in range(30):        # (warm: the first lattice of a process otherwise pays the GPU's clock ramp)
        f()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    sync()
    return (time.perf_counter() - t0) / n * 1e6


one = Lattice(T, L, L, L, kappa=kappa, mu=mu)
prep(one, syn.gauge_field(1, T, L, L, L), 1, 0)
print("%dx%d^3 unsplit      : sw_term %.0f us   sw_all %.0f us" % (T, L, timed(lambda: one.sw_term(None, kappa, c_sw), one.sync), timed(lambda: one.sw_all(kappa, c_sw), one.sync)), flush=True)
one.close()
world = 2
lats = [Lattice(T, L, L, L, kappa=kappa, mu=mu, nproc_t=world, proc_t=r) for r in range(world)]
for r, lat in enumerate(lats):
    prep(lat, syn.gauge_field(1, T, L, L, L, world, r), world, r)


def sync_all():
    for lat in lats:
        lat.sync()


t_term = timed(lambda: lats[0].sw_term(None, kappa, c_sw), sync_all)
t_all = timed(lambda: multi_sw_all(lats, kappa, c_sw), sync_all)
print("%dx%d^3 rank of a split: sw_term %.0f us   sw_all (both contexts, incl. the halo copies) %.0f us = %.0f per rank" % (T, L, t_term, t_all, t_all / world), flush=True)
for lat in lats:
    lat.close()
