"""two-process probe of the direct carrier: the bench.py headline flow step by step.  usage: r04_mp_probe.py RANK WORLD JOB [T L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice
from tmlqcd_amd import synthetic as syn
rank, world, job = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
T, L = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (16, 16)
for pre in range(0):   # two lattices before: created, used over the direct carrier, closed (bench.py's configs[3] and strong_32 legs)
    Tp = 32 if pre == 0 else 8
    lp = Lattice(Tp, L, L, L, kappa=0.125, mu=0.01, nproc_t=world, proc_t=rank, device=0)
    lp.comm_init_shm(job + "_pre%d" % pre); lp.comm_init_ipc()
    lp.set_gauge(syn.gauge_field(7, Tp, L, L, L, world, rank))
    k = lp.field(syn.spinor_field_eo(8, 0, Tp, L, L, L, world, rank)); l, q, Pp = lp.field(), lp.field(), lp.field()
    lp.Hopping_Matrix(0, l, k); lp.Qtm_pm_psi(q, k); lp.square_norm(q, lp.Vh, 1); lp.cg_her(Pp, k, 2000, 1e-20, 1, lp.Vh)
    lp.bench_hopping(k, l, q, 22)
    for _ in range(22):
        lp.Hopping_Matrix_nocom(0, l, k); lp.Hopping_Matrix_nocom(1, q, l)
    lp.cg_her(Pp, k, 5, 0.0, 1, lp.Vh); lp.cg_her(Pp, k, 25, 0.0, 1, lp.Vh)
    lp.sync(); lp.close()
    print("rank %d: pre-lattice %d done" % (rank, pre), flush=True)
lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01, nproc_t=world, proc_t=rank, device=0)
lat.comm_init_shm(job)
lat.comm_init_ipc()
lat.set_gauge(syn.gauge_field(7, T, L, L, L, world, rank))
f0 = lat.field(syn.spinor_field_eo(8, 0, T, L, L, L, world, rank)); f1, f2 = lat.field(), lat.field()
P, Q = lat.field(), lat.field(syn.spinor_field_eo(9, 1, T, L, L, L, world, rank))
def step(name, fn):
    t0 = time.time()
    try:
        r = fn(); lat.sync()
        print("rank %d: %-28s ok  %.2f s %s" % (rank, name, time.time() - t0, r if isinstance(r, (int, float, tuple)) else ""), flush=True)
    except Exception as e:
        print("rank %d: %-28s FAILED after %.2f s: %r" % (rank, name, time.time() - t0, e), flush=True)
        sys.exit(1)
step("bench_hopping 22", lambda: lat.bench_hopping(f0, f1, f2, 22))
for n in (5, 25, 5, 25):
    step("zero", P.zero)
    step("cg_her %d" % n, lambda: lat.cg_her(P, Q, n, 0.0, 1, lat.Vh)[0])
def herm():
    src = syn.spinor_field_eo(8, 0, T, L, L, L, world, rank)
    y, a, b = lat.field(src), lat.field(), lat.field()
    lat.op("Qtm_plus_psi", a, Q)
    lat.op("Qtm_minus_psi", b, y)
    s1, s2 = lat.scalar_prod_r(y, a, lat.Vh, 1), lat.scalar_prod_r(b, Q, lat.Vh, 1)
    if "nofree" not in sys.argv:
        for f in (y, a, b):
            f.free()
    return abs(s1 - s2) / max(abs(s1), 1e-300)
step("hermiticity", herm)
step("mixed_cg_her 2", lambda: lat.mixed_cg_her(P, Q, 2, 1e-20, 1, lat.Vh))
step("cg_her full", lambda: lat.cg_her(P, Q, 5000, 1e-20, 1, lat.Vh)[0])
step("mixed_cg_her full", lambda: lat.mixed_cg_her(P, Q, 5000, 1e-20, 1, lat.Vh))
lat.close()
