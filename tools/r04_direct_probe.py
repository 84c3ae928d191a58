"""Where does the one-kernel form of the direct carrier lose its time?  Per-stencil times (HIP events) of
  plain : Hopping_Matrix(0, f1, f0); Hopping_Matrix(1, f2, f1)   -- every stencil packs, waits; nobody pushes ahead
  bench : tmhip_bench_hopping                                    -- first stencil packs + pushes ahead (HOP_FEED), second is chained
Usage: r04_direct_probe.py T [loopback] [name=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

T, L = int(sys.argv[1]), 32
lb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
f0 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L))
f1, f2 = lat.field(), lat.field()
for kv in sys.argv[3:]:
    lat.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if lb:
    lat.set_loopback(lb)
n = 200
for what in ("plain", "bench", "nocom"):
    for rep in range(2):
        lat.sync()
        lat.event_record(10)
        if what == "plain":
            for _ in range(n):
                lat.Hopping_Matrix(0, f1, f0)
                lat.Hopping_Matrix(1, f2, f1)
        elif what == "nocom":
            for _ in range(n):
                lat.Hopping_Matrix_nocom(0, f1, f0)
                lat.Hopping_Matrix_nocom(1, f2, f1)
        else:
            lat.bench_hopping(f0, f1, f2, n)
        lat.event_record(11)
        ms = lat.event_elapsed_ms(10, 11)
    print("T=%d loopback %d %s %-6s %.2f us per stencil" % (T, lb, " ".join(sys.argv[3:]), what, 1e3 * ms / (2 * n)), flush=True)
lat.sync()
lat.close()
