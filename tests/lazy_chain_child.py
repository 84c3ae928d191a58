"""Child process of tests/test_gpu_lazy.py::test_faults_of_the_host_program_reach_its_own_handler: a host program whose own SIGSEGV
handler is installed BEFORE the library's.  Prints one line, "OK ..." or the reason it is not."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import TOL, random_gauge, random_spinor, rel_err  # noqa: E402

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

T = L = 8
V = T * L ** 3
N = V // 2
g = random_gauge(5, V)
C.memmove(stub.stub_init(T, L, L, L), g.ctypes.data_as(VP), g.nbytes)
stub.stub_boundary(0.13, 0.0, 0.0, 0.0, 0.0)
stub.stub_set_mu(0.01)
assert stub.stub_install_segv_probe() == 0                    # the program's handler first ...
assert stub.stub_touch_guard() == 1                           # ... and it works on its own
blk = np.zeros(3 * N * 24 + 64)                               # (an mmap of its own: 3 fields back to back)
f = [blk[8 + i * N * 24: 8 + (i + 1) * N * 24].reshape(N, 4, 3, 2) for i in range(3)]
f[0][:] = random_spinor(6, N)


def p(a):
    return a.ctypes.data_as(VP)


d.tmlqcd_hip_set_residency(0)
d.Hopping_Matrix(0, p(f[2]), p(f[0]))                         # coherent mode: the reference
want = f[2].copy()
d.tmlqcd_hip_set_residency(2)                                 # lazy: the library's handler goes on top, chaining to the program's
d.Hopping_Matrix(0, p(f[1]), p(f[0]))
if stub.stub_touch_guard() != 2:
    print("the program's handler did not see its fault while fields were stale"); sys.exit(1)
if not np.array_equal(f[1], want):                            # the library's faults, served as before
    print("lazy result differs", rel_err(f[1], want)); sys.exit(1)
f[0][3] *= 2.0                                                # a store into the write-protected input
d.Hopping_Matrix(0, p(f[1]), p(f[0]))
d.tmlqcd_hip_set_residency(0)
d.Hopping_Matrix(0, p(f[2]), p(f[0]))
if stub.stub_touch_guard() != 3 or not np.array_equal(f[1], f[2]):
    print("second round failed"); sys.exit(1)
st = (C.c_ulong * 4)(); d.tmlqcd_hip_lazy_stats(st)
d.tmlqcd_hip_finalize()
print("OK program faults 3, library faults %d stores %d" % (st[0], st[3]))
