"""Child process of tests/test_gpu_lazy.py::test_arrays_that_cannot_be_watched: a host program whose spinor arrays live where the lazy
mode must not take pages away -- inside the malloc heap, inside a thread's malloc arena, in a shared mapping.  `python
lazy_unwatchable_child.py heap|arena|shared`: the arrays are recognised (TMLQCD_HIP_LAZY_DEBUG says why), left unwatched, copied per
call, and the results are the coherent mode's.  With TMLQCD_HIP_LAZY_FORCE_WATCH=1 (test hook) they ARE watched: the first fault on one
of them must end the process with a message -- not hang it, which is what a fault taken inside malloc did in round 3."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import random_gauge, random_spinor  # noqa: E402

where = sys.argv[1]
VP = C.c_void_p
d0 = os.path.join(ROOT, "tests", "host_stub")
so, src = os.path.join(d0, "libtmhost.so"), os.path.join(d0, "globals.c")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-o", so, src, "-lm"])
stub = C.CDLL(so, mode=C.RTLD_GLOBAL)
import tmlqcd_amd  # noqa: E402
tmlqcd_amd.load_library()
d = C.CDLL(os.path.join(ROOT, "tmlqcd_amd", "lib", "libtmlqcd_dropin.so"), mode=C.RTLD_GLOBAL)
stub.stub_init.restype = VP; stub.stub_init.argtypes = [C.c_int] * 4
stub.stub_boundary.argtypes = [C.c_double] * 5
stub.stub_set_mu.argtypes = [C.c_double]
d.Hopping_Matrix.argtypes = [C.c_int, VP, VP]
d.tmlqcd_hip_set_residency.argtypes = [C.c_int]
libc = C.CDLL(None, use_errno=True)
libc.malloc.restype = VP; libc.malloc.argtypes = [C.c_size_t]
libc.mmap.restype = VP; libc.mmap.argtypes = [VP, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.mallopt.argtypes = [C.c_int, C.c_int]

T = L = 8
V = T * L ** 3
N = V // 2
g = random_gauge(5, V)
C.memmove(stub.stub_init(T, L, L, L), g.ctypes.data_as(VP), g.nbytes)
stub.stub_boundary(0.13, 0.0, 0.0, 0.0, 0.0)
stub.stub_set_mu(0.01)
nb = 3 * N * 192 + 64
if where == "shared":
    base = libc.mmap(None, (nb + 4095) // 4096 * 4096, 3, 0x21, -1, 0)          # MAP_SHARED | MAP_ANONYMOUS
elif where == "heap":
    libc.mallopt(-3, 1 << 30)                                                    # M_MMAP_THRESHOLD: the PROGRAM keeps big blocks in its heap
    junk = [libc.malloc(1 << 20) for _ in range(4)]
    base = libc.malloc(nb)
else:                                                                            # a thread's arena
    libc.mallopt(-3, 1 << 30)
    box = {}
    th = threading.Thread(target=lambda: box.update(p=libc.malloc(nb)))
    th.start(); th.join()
    base = box["p"]
assert base not in (None, 0, C.c_void_p(-1).value)
blk = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_double)), shape=(nb // 8,))
f = [blk[8 + i * N * 24: 8 + (i + 1) * N * 24].reshape(N, 4, 3, 2) for i in range(3)]
f[0][:] = random_spinor(6, N)


def p(a):
    return a.ctypes.data_as(VP)


d.tmlqcd_hip_set_residency(0)
d.Hopping_Matrix(0, p(f[2]), p(f[0]))                         # coherent mode: the reference
want = f[2].copy()
d.tmlqcd_hip_set_residency(2)                                 # lazy
d.Hopping_Matrix(0, p(f[1]), p(f[0]))
got = f[1].copy()                                             # FORCE_WATCH: this load faults on a page that must not be watched -> message + exit
f[0][3] *= 2.0
d.Hopping_Matrix(0, p(f[1]), p(f[0]))
d.tmlqcd_hip_set_residency(0)
d.Hopping_Matrix(0, p(f[2]), p(f[0]))
if not np.array_equal(got, want) or not np.array_equal(f[1], f[2]):
    print("results differ from the coherent mode's"); sys.exit(2)
st = (C.c_ulong * 4)(); d.tmlqcd_hip_lazy_stats(st)
d.tmlqcd_hip_finalize()
print("OK %s: faults served %d" % (where, st[0]))
