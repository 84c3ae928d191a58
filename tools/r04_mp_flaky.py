"""two-process flakiness probe of the direct carrier: Q_plus = Q_minus^dagger many times.  usage: r04_mp_flaky.py RANK WORLD JOB [name=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice
from tmlqcd_amd import synthetic as syn
rank, world, job = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
T, L = 16, 16
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01, nproc_t=world, proc_t=rank, device=0)
for kv in sys.argv[4:]:
    if kv.split("=")[0] != "faces":
        lat.set_option(kv.split("=")[0], int(kv.split("=")[1]))
lat.comm_init_shm(job)
if "faces=ring" not in sys.argv:
    lat.comm_init_ipc()
lat.set_gauge(syn.gauge_field(7, T, L, L, L, world, rank))
Q = lat.field(syn.spinor_field_eo(9, 1, T, L, L, L, world, rank))
src = syn.spinor_field_eo(8, 0, T, L, L, L, world, rank)
bad = 0
for it in range(60):
    y, a, b = lat.field(src), lat.field(), lat.field()
    lat.op("Qtm_plus_psi", a, Q)
    lat.op("Qtm_minus_psi", b, y)
    s1, s2 = lat.scalar_prod_r(y, a, lat.Vh, 1), lat.scalar_prod_r(b, Q, lat.Vh, 1)
    dev = abs(s1 - s2) / max(abs(s1), 1e-300)
    if dev > 1e-12:
        bad += 1
        print("rank %d it %d dev %.3e" % (rank, it, dev), flush=True)
    if it % 2:
        for f in (y, a, b):
            f.free()
lat.sync()
print("rank %d %s: %d bad of 60" % (rank, " ".join(sys.argv[4:]), bad), flush=True)
lat.close()
