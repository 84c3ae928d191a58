"""TEST INFRASTRUCTURE ONLY.  ctypes binding of the ILDG part of oracle/libtmoracle.so (oracle/ildg_oracle.c) and, where it was
built (oracle/_ref/libtmref_dml.so: the reference's io/dml.c + io/DML_crc32.c compiled in place), of the reference's checksum code.
Gauge fields: float64 [V][4][3][3][2] in the layout of g_gauge_field (lexicographic site, mu = t, x, y, z)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(os.path.join(_HERE, "libtmoracle.so"))
        vp, i, u = C.c_void_p, C.c_int, C.c_uint32
        lib.tmo_crc32.restype = u
        lib.tmo_crc32.argtypes = [u, vp, C.c_size_t]
        lib.tmo_checksum_accum.argtypes = [vp, u, vp, C.c_size_t]
        lib.tmo_ildg_unpack.argtypes = [vp, i, i, i, i, i, u, vp, vp]
        lib.tmo_ildg_pack.argtypes = [vp, i, i, i, i, i, u, vp, vp]
        lib.tmo_write_gauge_field.argtypes = [C.c_char_p, i, i, i, i, i, vp, C.c_char_p, vp]
        lib.tmo_write_gauge_field.restype = i
        lib.tmo_read_gauge_field.argtypes = [C.c_char_p, i, i, i, i, i, vp, vp]
        lib.tmo_read_gauge_field.restype = i
        _LIB = lib
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def crc32(buf, crc=0):
    b = np.ascontiguousarray(np.frombuffer(bytes(buf), dtype=np.uint8))
    return int(_lib().tmo_crc32(crc, _p(b), b.size))


def checksum_accum(sums, rank, buf):
    b = np.ascontiguousarray(np.frombuffer(bytes(buf), dtype=np.uint8))
    s = np.array(sums, dtype=np.uint32)
    _lib().tmo_checksum_accum(_p(s), rank, _p(b), b.size)
    return int(s[0]), int(s[1])


def unpack(file_bytes, prec, T, LX, LY, LZ, rank0=0):
    b = np.ascontiguousarray(np.frombuffer(file_bytes, dtype=np.uint8) if not isinstance(file_bytes, np.ndarray) else file_bytes)
    gf = np.zeros((T * LX * LY * LZ, 4, 3, 3, 2))
    s = np.zeros(2, dtype=np.uint32)
    _lib().tmo_ildg_unpack(_p(b), prec, T, LX, LY, LZ, rank0, _p(gf), _p(s))
    return gf, (int(s[0]), int(s[1]))


def pack(gf, prec, T, LX, LY, LZ, rank0=0):
    g = np.ascontiguousarray(gf[:T * LX * LY * LZ], dtype=np.float64)
    out = np.zeros(T * LX * LY * LZ * (576 if prec == 64 else 288), dtype=np.uint8)
    s = np.zeros(2, dtype=np.uint32)
    _lib().tmo_ildg_pack(_p(out), prec, T, LX, LY, LZ, rank0, _p(g), _p(s))
    return out, (int(s[0]), int(s[1]))


def write_gauge_field(filename, gf, prec, T, LX, LY, LZ, xlf=None):
    g = np.ascontiguousarray(gf[:T * LX * LY * LZ], dtype=np.float64)
    s = np.zeros(2, dtype=np.uint32)
    rc = _lib().tmo_write_gauge_field(str(filename).encode(), prec, T, LX, LY, LZ, _p(g), xlf.encode() if xlf else None, _p(s))
    return rc, (int(s[0]), int(s[1]))


def read_gauge_field(filename, prec, T, LX, LY, LZ):
    gf = np.zeros((T * LX * LY * LZ, 4, 3, 3, 2))
    s = np.zeros(4, dtype=np.uint32)
    rc = _lib().tmo_read_gauge_field(str(filename).encode(), prec, T, LX, LY, LZ, _p(gf), _p(s))
    return rc, gf, tuple(int(x) for x in s)


# ---- the reference's own checksum objects (oracle/_ref/libtmref_dml.so), when built
def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libtmref_dml.so"))


def _ref():
    global _REF
    if _REF is None:
        lib = C.CDLL(os.path.join(_HERE, "_ref", "libtmref_dml.so"))
        lib.DML_crc32.restype = C.c_uint32
        lib.DML_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        lib.DML_checksum_accum.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
        _REF = lib
    return _REF


def ref_crc32(buf, crc=0):
    b = np.ascontiguousarray(np.frombuffer(bytes(buf), dtype=np.uint8))
    return int(_ref().DML_crc32(crc, _p(b), b.size))


def ref_checksum(record, site_bytes, rank0=0):
    """DML_checksum_accum (io/dml.c:49) over the sites of a binary record, rank = rank0 + site number, as the reader calls it."""
    b = np.ascontiguousarray(np.frombuffer(record, dtype=np.uint8) if not isinstance(record, np.ndarray) else record)
    s = np.zeros(2, dtype=np.uint32)
    for f in range(b.size // site_bytes):
        chunk = np.ascontiguousarray(b[f * site_bytes:(f + 1) * site_bytes])
        _ref().DML_checksum_accum(_p(s), rank0 + f, _p(chunk), site_bytes)
    return int(s[0]), int(s[1])
