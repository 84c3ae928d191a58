"""tmlqcd_amd -- MI355X-native (gfx950) implementation of tmLQCD's Wilson twisted-mass hot path.

The product is the C-ABI shared library `tmlqcd_amd/lib/libtmlqcd_hip.so` (hand-written HIP,
include/tmlqcd_hip.h) plus `libtmlqcd_dropin.so` (the reference's own symbol names,
include/tmlqcd_dropin.h).  This Python package is only a thin ctypes mirror of that ABI so the
parity tests and bench.py can drive it; there is no CPU fallback: importing `tmlqcd_amd.hip`
raises if the HIP library has not been built.
"""
from .hip import Lattice, Field, load_library, library_path, EO, OE  # noqa: F401
