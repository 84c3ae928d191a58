"""cg_her on small lattices: the iteration that adds up its partial sums inside the residual stencil and the (P, p) kernel ("cg_self" 1,
default) against the one with two sum + scalar kernels (0): iterations per second on a live residual (the method of bench.py's cg_16
leg), iterations to 1e-10 and the largest deviation between the two solutions.  Usage: cg_self_ab.py [L ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402


def rate(lat, P, Q, reps=20, n_short=5, n_long=25):
    def solve(n):
        P.zero(); lat.sync()
        t0 = time.perf_counter()
        lat.cg_her(P, Q, n, 0.0, 1, lat.Vh)
        lat.sync()
        return time.perf_counter() - t0
    solve(n_short); solve(n_long)
    ts = tl = 0.0
    for _ in range(reps):
        ts += solve(n_short); tl += solve(n_long)
    return reps * (n_long - n_short) / (tl - ts)


for L in [int(a) for a in sys.argv[1:]] or [8, 12, 16]:
    lat = Lattice(L, L, L, L, kappa=0.125, mu=0.01)
    lat.set_gauge(syn.gauge_field(7, L, L, L, L))
    P, Q = lat.field(), lat.field(syn.spinor_field_eo(9, 1, L, L, L, L))
    res = {0: [], 1: []}
    for rnd in range(3):
        for v in (0, 1):
            lat.set_option("cg_self", v)
            res[v].append(rate(lat, P, Q))
    its, sol = {}, {}
    for v in (0, 1):
        lat.set_option("cg_self", v)
        P.zero()
        its[v] = lat.cg_her(P, Q, 2000, 1e-20, 1, lat.Vh)[0]
        sol[v] = P.download()
    dev = np.abs(sol[0] - sol[1]).max() / np.abs(sol[0]).max()
    print("L=%2d cg_self 0: %s it/s   cg_self 1: %s it/s   iterations to 1e-10: %d / %d   max rel dev of the solutions %.1e"
          % (L, " ".join("%.0f" % x for x in res[0]), " ".join("%.0f" % x for x in res[1]), its[0], its[1], dev), flush=True)
    lat.close()
