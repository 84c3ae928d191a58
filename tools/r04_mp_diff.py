"""two-process differential probe: the same operation sequence on a ring-transport lattice and on a direct-carrier lattice of the same
process, compared after every operation.  usage: r04_mp_diff.py RANK WORLD JOB [name=value ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmlqcd_amd import Lattice
from tmlqcd_amd import synthetic as syn
rank, world, job = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
T, L = 16, 16
def make(direct):
    lat = Lattice(T, L, L, L, kappa=0.125, mu=0.01, nproc_t=world, proc_t=rank, device=0)
    for kv in sys.argv[4:]:
        lat.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    lat.comm_init_shm(job + ("d" if direct else "r"))
    if direct:
        lat.comm_init_ipc()
    lat.set_gauge(syn.gauge_field(7, T, L, L, L, world, rank))
    return lat
A, B = make(False), make(True)
src0 = syn.spinor_field_eo(8, 0, T, L, L, L, world, rank); src1 = syn.spinor_field_eo(9, 1, T, L, L, L, world, rank)
FA = dict(f0=A.field(src0), f1=A.field(), f2=A.field(), P=A.field(), Q=A.field(src1), a=A.field(), b=A.field())
FB = dict(f0=B.field(src0), f1=B.field(), f2=B.field(), P=B.field(), Q=B.field(src1), a=B.field(), b=B.field())
def both(name, fn, outs):
    ra, rb = fn(A, FA), fn(B, FB)
    A.sync(); B.sync()
    for o in outs:
        x, y = FA[o].download(), FB[o].download()
        dev = float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-300))
        if not dev < 1e-12:
            print("rank %d: %s: field %s differs by %.3e (returns %r / %r)" % (rank, name, o, dev, ra, rb), flush=True)
            return False
    return True
ok = True
for rep in range(6):
    seq = [("bench %d" % rep, lambda l, F: l.bench_hopping(F["f0"], F["f1"], F["f2"], 22), ["f1", "f2"])]
    for n in (5, 25):
        seq.append(("zero", lambda l, F: F["P"].zero(), []))
        seq.append(("cg %d" % n, lambda l, F, n=n: l.cg_her(F["P"], F["Q"], n, 0.0, 1, l.Vh)[0], ["P"]))
    seq.append(("Qtm_plus", lambda l, F: l.op("Qtm_plus_psi", F["a"], F["Q"]), ["a"]))
    seq.append(("Qtm_minus", lambda l, F: l.op("Qtm_minus_psi", F["b"], F["f0"]), ["b"]))
    seq.append(("mixed", lambda l, F: l.mixed_cg_her(F["P"], F["Q"], 2, 1e-20, 1, l.Vh), []))
    for name, fn, outs in seq:
        if not both("%s (rep %d)" % (name, rep), fn, outs):
            ok = False
            break
    if not ok:
        break
print("rank %d: %s" % (rank, "all equal" if ok else "MISMATCH"), flush=True)
A.close(); B.close()
