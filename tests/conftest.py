import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
    import tmlqcd_amd
    tmlqcd_amd.load_library()
    dropin = C.CDLL(os.path.join(ROOT, "tmlqcd_amd", "lib", "libtmlqcd_dropin.so"), mode=C.RTLD_GLOBAL)
    return stub, dropin
