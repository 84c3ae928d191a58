"""One rank of tests/test_gpu_multiprocess.py::test_drop_in_symbols_on_t_split_ranks: `python mp_dropin_worker.py RANK WORLD JOB OUTDIR`.
A host program on a T-split lattice the way an MPI tmLQCD run is one (g_nproc_t, g_proc_coords, RAND = the two halo slices of the gauge
field): it calls the REFERENCE-NAMED symbols of libtmlqcd_dropin.so -- Hopping_Matrix, Qtm_pm_psi, square_norm / scalar_prod_r with
parallel = 1, cg_her with a function pointer -- after tmlqcd_hip_comm_init_shm.  WORLD = 1: the unsplit host program."""
import ctypes as C
import os
import subprocess
import sys

import faulthandler

import numpy as np

faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("MP_WORKER_TIMEOUT", "240")), exit=True)     # a hung rank says where, and ends

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tmlqcd_amd import synthetic as syn  # noqa: E402

rank, world, job, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
VP = C.c_void_p
d0 = os.path.join(ROOT, "tests", "host_stub")
so, src = os.path.join(d0, "libtmhost.so"), os.path.join(d0, "globals.c")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-o", so + ".%d" % os.getpid(), src, "-lm"])
    os.replace(so + ".%d" % os.getpid(), so)          # (several ranks may get here at once)
stub = C.CDLL(so, mode=C.RTLD_GLOBAL)
import tmlqcd_amd  # noqa: E402
tmlqcd_amd.load_library()
d = C.CDLL(os.path.join(ROOT, "tmlqcd_amd", "lib", "libtmlqcd_dropin.so"), mode=C.RTLD_GLOBAL)
stub.stub_init_rank.restype = VP; stub.stub_init_rank.argtypes = [C.c_int] * 6
stub.stub_boundary.argtypes = [C.c_double] * 5
stub.stub_set_mu.argtypes = [C.c_double]
d.Hopping_Matrix.argtypes = [C.c_int, VP, VP]
d.Qtm_pm_psi.argtypes = [VP, VP]
d.square_norm.restype = C.c_double; d.square_norm.argtypes = [VP, C.c_int, C.c_int]
d.scalar_prod_r.restype = C.c_double; d.scalar_prod_r.argtypes = [VP, VP, C.c_int, C.c_int]
d.cg_her.restype = C.c_int; d.cg_her.argtypes = [VP, VP, C.c_int, C.c_double, C.c_int, C.c_int, VP]
d.tmlqcd_hip_comm_init_shm.argtypes = [C.c_char_p]
d.tmlqcd_hip_set_residency.argtypes = [C.c_int]

Tg, L = 8, 8
T = Tg // world
N = T * L ** 3 // 2
g = syn.gauge_field(41, T, L, L, L, world, rank)                       # [VOLUMEPLUSRAND][4] su3, halo slices filled as xchange_gauge would
C.memmove(stub.stub_init_rank(T, L, L, L, world, rank), g.ctypes.data_as(VP), g.nbytes)
stub.stub_boundary(0.13, 1.0, 0.0, 0.0, 0.0)
stub.stub_set_mu(0.02)
if world > 1:
    d.tmlqcd_hip_comm_init_shm(job.encode())
    if os.environ.get("MP_FACES") == "direct":
        assert d.tmlqcd_hip_comm_init_ipc() == 0


def p(a):
    return a.ctypes.data_as(VP)


res = {}
modes = [m for m in ((0, "coherent"), (2, "lazy")) if str(m[0]) in os.environ.get("MP_DROPIN_MODES", "02")]
for mode, tag in modes:
    d.tmlqcd_hip_set_residency(mode)
    sys.stderr.write("rank %d: %s\n" % (rank, tag)); sys.stderr.flush()
    k = syn.spinor_field_eo(42, 0, T, L, L, L, world, rank)
    b = syn.spinor_field_eo(43, 1, T, L, L, L, world, rank)
    l, x = np.zeros_like(k), np.zeros_like(k)
    d.Hopping_Matrix(1, p(l), p(k)); res[tag + "_hop"] = l.copy()
    d.Qtm_pm_psi(p(l), p(k)); res[tag + "_qtm"] = l.copy()
    res[tag + "_sums"] = np.array([d.square_norm(p(l), N, 1), d.scalar_prod_r(p(l), p(k), N, 1), d.square_norm(p(l), N, 0)])
    it = d.cg_her(p(x), p(b), 2000, 1e-18, 1, N, C.cast(d.Qtm_pm_psi, VP))
    res[tag + "_cg"] = x.copy(); res[tag + "_it"] = np.array([it])
d.tmlqcd_hip_set_residency(0)
d.tmlqcd_hip_finalize()
np.savez(os.path.join(outdir, "dropin_%d_of_%d.npz" % (rank, world)), **res)
print("rank %d of %d done" % (rank, world), flush=True)
