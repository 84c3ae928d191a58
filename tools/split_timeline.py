"""Run a few split-phase (loopback) stencils for a rocprofv3 kernel trace.  Usage: split_timeline.py L T [nocom|unsplit] [loopback]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

L, T = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "comm"
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01)
lat.set_gauge(syn.gauge_field(1, T, L, L, L))
f0 = lat.field(syn.spinor_field_eo(2, 0, T, L, L, L))
f1, f2 = lat.field(), lat.field()
for kv in os.environ.get("TL_OPTS", "").split(","):      # e.g. TL_OPTS=split_sync=1
    if "=" in kv:
        lat.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if mode != "unsplit":
    lat.set_loopback(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
if mode == "nocom":
    for _ in range(20):
        lat.Hopping_Matrix_nocom(0, f1, f0)
        lat.Hopping_Matrix_nocom(1, f2, f1)
else:
    lat.bench_hopping(f0, f1, f2, 20)
lat.sync()
lat.close()
