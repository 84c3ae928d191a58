import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Device-side waits of the split path are bounded (tmlqcd_hip.h "flag_timeout_ms", default 120 s).  In the test suite a wait that
# long can only be a bug (a deadlock between the stencil kernel and the exchange it waits for): let it surface as an error within
# 20 s instead of two minutes per call.  The late-neighbour tests delay by 6 s, well inside.
os.environ.setdefault("TMLQCD_HIP_FLAG_TIMEOUT_S", "20")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (and, where hipcc exists, the HIP library) are built."""
    import __graft_entry__ as ge
    ge.build(quiet=True)


@pytest.fixture(scope="session")
def host_stub():
    """(host-program stub, drop-in library) loaded the way a tmLQCD executable would link them."""
    import ctypes as C
    import subprocess
    d = os.path.join(ROOT, "tests", "host_stub")
    so = os.path.join(d, "libtmhost.so")
    src = os.path.join(d, "globals.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-fPIC", "-shared", "-o", so, src, "-lm"])
    stub = C.CDLL(so, mode=C.RTLD_GLOBAL)
    stub.stub_init.restype = C.c_void_p
    stub.stub_init.argtypes = [C.c_int] * 4
    stub.stub_boundary.argtypes = [C.c_double] * 5
    stub.stub_set_mu.argtypes = [C.c_double]
    stub.stub_get_mu.restype = C.c_double
    stub.stub_set_csw.argtypes = [C.c_double]
    stub.stub_set_mu3.argtypes = [C.c_double]
    import tmlqcd_amd
    tmlqcd_amd.load_library()
    dropin = C.CDLL(os.path.join(ROOT, "tmlqcd_amd", "lib", "libtmlqcd_dropin.so"), mode=C.RTLD_GLOBAL)
    return stub, dropin


@pytest.fixture(scope="session")
def c_host_program():
    """tests/c_host/mini_benchmark: a C main in the shape of benchmark.c, linked at LINK TIME (not dlopen) against
    libtmlqcd_dropin.so + libtmlqcd_hip.so the way INTEGRATION.md §2 puts them on tmLQCD's link line."""
    import subprocess
    d = os.path.join(ROOT, "tests", "c_host")
    exe = os.path.join(d, "mini_benchmark")
    srcs = [os.path.join(d, "mini_benchmark.c"), os.path.join(ROOT, "tests", "host_stub", "globals.c")]
    libs = [os.path.join(ROOT, "tmlqcd_amd", "lib", "libtmlqcd_dropin.so"), os.path.join(ROOT, "oracle", "libtmoracle.so")]
    newest = max(os.path.getmtime(p) for p in srcs + libs)
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-Wall", "-o", exe] + srcs +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
                               "-L" + os.path.join(ROOT, "tmlqcd_amd", "lib"), "-ltmlqcd_dropin", "-ltmlqcd_hip",
                               "-L" + os.path.join(ROOT, "oracle"), "-ltmoracle", "-lm",
                               "-Wl,-rpath,$ORIGIN/../../tmlqcd_amd/lib", "-Wl,-rpath,$ORIGIN/../../oracle", "-Wl,-rpath,/opt/rocm/lib"])
    return exe
