"""GPU: ILDG gauge configurations through the HIP path (ildg.hip) -- unpack / pack kernels, LIME reader / writer, drop-in
read_gauge_field / write_gauge_field -- bit for bit against the restatement (oracle/ildg_oracle.c) and the committed fixtures,
whose checksums come from the reference's own io/dml.c (tests/test_ildg_oracle.py, oracle/make_golden_ildg.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import ildgbind as ib
from tmlqcd_amd import synthetic as syn
from tests.util import random_spinor, rel_err, TOL

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLD, "ildg_checksums.json")))
T, LX, LY, LZ = META["lattice"]


@pytest.mark.parametrize("prec", [64, 32])
def test_read_the_fixture_files(prec):
    from tmlqcd_amd import Lattice
    name = "ildg_%dx%dx%dx%d_prec%d.lime" % (T, LX, LY, LZ, prec)
    meta = META["files"][name]
    lat = Lattice(T, LX, LY, LZ, kappa=0.125, mu=0.01)
    rc, gf, info = lat.read_gauge_field(os.path.join(GOLD, name), prec=prec)
    assert rc == 0 and info.gauge_read == 1
    assert "%08x" % info.suma == meta["suma"] and "%08x" % info.sumb == meta["sumb"]      # == the reference's checksum code
    assert (info.suma_stored, info.sumb_stored) == (info.suma, info.sumb)
    assert (info.prec, info.lx, info.ly, info.lz, info.lt) == (prec, LX, LY, LZ, T)
    assert (info.xlf_info == META["xlf_info_text"].encode()) if prec == 64 else not info.xlf_info       # the plain text of io/utils_write_xlf.c:35-55
    _, want, _ = ib.read_gauge_field(os.path.join(GOLD, name), prec, T, LX, LY, LZ)
    assert np.array_equal(gf[:lat.V], want)                                               # the host's g_gauge_field
    assert np.array_equal(lat.gauge_download()[:lat.V], want)                             # the links resident in HBM
    # ... and the stencil's gauge copy was sorted from them: Hopping_Matrix as after set_gauge
    src = syn.spinor_field_eo(3, 1, T, LX, LY, LZ)
    k, l = lat.field(src), lat.field()
    lat.Hopping_Matrix(0, l, k)
    a = l.download()
    lat2 = Lattice(T, LX, LY, LZ, kappa=0.125, mu=0.01)
    lat2.set_gauge(want)
    k2, l2 = lat2.field(src), lat2.field()
    lat2.Hopping_Matrix(0, l2, k2)
    assert np.array_equal(a, l2.download())
    lat.close(); lat2.close()


@pytest.mark.parametrize("dims,prec", [((4, 4, 4, 4), 64), ((4, 4, 4, 4), 32), ((6, 10, 2, 12), 64), ((16, 16, 16, 16), 64), ((16, 16, 16, 16), 32)])
def test_unpack_and_pack_kernels_against_the_restatement(dims, prec):
    """Sizes with partial last blocks (V % 64 != 0) and a 16^4 record (38 MB); both directions, both precisions."""
    from tmlqcd_amd import Lattice
    Tt, X, Y, Z = dims
    g = syn.gauge_field(17, Tt, X, Y, Z)
    rec, sums = ib.pack(g, prec, Tt, X, Y, Z)
    lat = Lattice(Tt, X, Y, Z, kappa=0.125, mu=0.0)
    got = lat.gauge_unpack_ildg(rec, prec)
    assert got == sums
    want, _ = ib.unpack(rec, prec, Tt, X, Y, Z)
    assert np.array_equal(lat.gauge_download()[:lat.V], want)
    lat.set_gauge(g)
    out, s2 = lat.gauge_pack_ildg(prec)
    assert s2 == sums and np.array_equal(out, rec)
    lat.close()


def test_written_file_is_the_restatements_byte_for_byte(tmp_path):
    from tmlqcd_amd import Lattice
    g = syn.gauge_field(META["gauge_seed"], T, LX, LY, LZ)
    lat = Lattice(T, LX, LY, LZ, kappa=0.125, mu=0.0)
    lat.set_gauge(g)
    xlf = META["xlf_info_text"]
    for prec in (64, 32):
        p = tmp_path / ("w%d.lime" % prec)
        sums = lat.write_gauge_field(p, prec, xlf if prec == 64 else None)
        fixture = os.path.join(GOLD, "ildg_%dx%dx%dx%d_prec%d.lime" % (T, LX, LY, LZ, prec))
        assert p.read_bytes() == open(fixture, "rb").read()                               # records, padding, checksums: identical files
        assert "%08x" % sums[0] == META["files"][os.path.basename(fixture)]["suma"]
        rc, back, _ = ib.read_gauge_field(p, prec, T, LX, LY, LZ)
        assert rc == 0 and np.array_equal(back, g if prec == 64 else g.astype(np.float32).astype(np.float64))
    lat.close()


def test_reader_error_behaviour(tmp_path, capfd):
    """gauge_read.c:64-170: -1 with a message for a size / precision mismatch, a corrupted record, a missing checksum record;
    with the IO checks disabled (g_disable_IO_checks) the corrupted file is accepted."""
    from tmlqcd_amd import Lattice
    src = os.path.join(GOLD, "ildg_%dx%dx%dx%d_prec64.lime" % (T, LX, LY, LZ))
    raw = bytearray(open(src, "rb").read())
    lat = Lattice(T, LX, LY, LZ, kappa=0.125, mu=0.0)
    assert lat.read_gauge_field(src, prec=32)[0] == -1
    assert "do not match those requested" in capfd.readouterr().err
    bad = bytearray(raw); bad[len(raw) // 2] ^= 0x01
    p = tmp_path / "flipped.lime"; p.write_bytes(bad)
    assert lat.read_gauge_field(p)[0] == -1
    assert "SciDAC checksum" in capfd.readouterr().err
    rc, _, info = lat.read_gauge_field(p, io_checks=False)
    assert rc == 0 and (info.suma, info.sumb) != (info.suma_stored, info.sumb_stored)
    from tests.test_ildg_oracle import records
    last = records(src)[-1]
    assert last[0] == "scidac-checksum"
    cut = raw[:len(raw) - 144 - (len(last[3]) + 7) // 8 * 8]             # drop the trailing scidac-checksum record
    p2 = tmp_path / "nochecksum.lime"; p2.write_bytes(cut)
    assert lat.read_gauge_field(p2)[0] == -1
    assert "scidac-checksum" in capfd.readouterr().err
    p3 = tmp_path / "notlime.lime"; p3.write_bytes(b"x" * 500)
    assert lat.read_gauge_field(p3)[0] == -1
    lat2 = Lattice(T, LX, LY, 2 * LZ, kappa=0.125, mu=0.0)
    assert lat2.read_gauge_field(src)[0] == -1
    lat.close(); lat2.close()


def test_t_split_ranks_pack_their_part_of_the_record():
    """Two contexts holding the two T-slabs: the record is the concatenation of their parts, the checksum the XOR (io/dml.c:63-66)."""
    from tmlqcd_amd import Lattice
    Tg, L, world = 8, 4, 2
    g = syn.gauge_field(23, Tg, L, L, L)
    rec, sums = ib.pack(g, 64, Tg, L, L, L)
    parts, xa, xb = [], 0, 0
    for r in range(world):
        lat = Lattice(Tg // world, L, L, L, kappa=0.125, mu=0.0, nproc_t=world, proc_t=r)
        lat.set_gauge(syn.gauge_field(23, Tg // world, L, L, L, world, r))
        out, s = lat.gauge_pack_ildg(64)
        parts.append(out); xa ^= s[0]; xb ^= s[1]
        lat.close()
    assert np.array_equal(np.concatenate(parts), rec) and (xa, xb) == sums


def test_reader_combines_checksums_through_the_reduction_communicator():
    """The multi-rank leg of the reader (own part of the record at its offset, ncclAllGather of the checksum words, XOR) with a
    one-rank RCCL communicator: the same calls a T-split rank makes."""
    from tmlqcd_amd import Lattice
    name = "ildg_%dx%dx%dx%d_prec64.lime" % (T, LX, LY, LZ)
    lat = Lattice(T, LX, LY, LZ, kappa=0.125, mu=0.01)
    lat.set_loopback(2)
    assert lat.comm_count() == (1, 1)
    rc, gf, info = lat.read_gauge_field(os.path.join(GOLD, name))
    assert rc == 0 and "%08x" % info.suma == META["files"][name]["suma"] and "%08x" % info.sumb == META["files"][name]["sumb"]
    assert np.array_equal(gf[:lat.V], syn.gauge_field(META["gauge_seed"], T, LX, LY, LZ))
    lat.close()


def test_drop_in_read_and_write_gauge_field(host_stub, tmp_path):
    """The reference's symbols on the host's g_gauge_field: read_gauge_field fills it, sets GaugeInfo and g_update_gauge_copy, and the
    next Hopping_Matrix runs on the links the reader left in HBM; write_gauge_field writes the fixture back byte for byte."""
    from oracle.oraclebind import Oracle
    stub, d = host_stub
    VP = C.c_void_p
    V = T * LX * LY * LZ
    gptr = stub.stub_init(T, LX, LY, LZ)
    stub.stub_boundary(0.125, 0.0, 0.0, 0.0, 0.0)
    stub.stub_set_mu(0.01)
    stub.stub_set_io.argtypes = [C.c_int, C.c_int]
    stub.stub_set_io(64, 0)
    d.read_gauge_field.restype = C.c_int; d.read_gauge_field.argtypes = [C.c_char_p, VP]
    d.write_gauge_field.restype = C.c_int; d.write_gauge_field.argtypes = [C.c_char_p, C.c_int, VP]
    d.Hopping_Matrix.argtypes = [C.c_int, VP, VP]
    gfpp = C.c_void_p.in_dll(stub, "g_gauge_field")
    fixture = os.path.join(GOLD, "ildg_%dx%dx%dx%d_prec64.lime" % (T, LX, LY, LZ))
    assert d.read_gauge_field(fixture.encode(), gfpp) == 0
    assert stub.stub_gauge_flag() == 1
    host = np.ctypeslib.as_array(C.cast(gptr, C.POINTER(C.c_double)), shape=(V, 4, 3, 3, 2)).copy()
    g = syn.gauge_field(META["gauge_seed"], T, LX, LY, LZ)
    assert np.array_equal(host, g)

    class GI(C.Structure):
        _fields_ = [("plaq", C.c_double), ("gaugeRead", C.c_int), ("suma", C.c_uint), ("sumb", C.c_uint), ("xlf", C.c_char_p), ("lfn", C.c_char_p)]
    gi = GI.in_dll(d, "GaugeInfo")
    assert gi.gaugeRead == 1 and "%08x" % gi.suma == META["files"][os.path.basename(fixture)]["suma"] and gi.xlf == META["xlf_info_text"].encode()
    orc = Oracle(T, LX, LY, LZ, kappa=0.125, mu=0.01, theta=(0, 0, 0, 0), threads=2)
    orc.set_gauge(g)
    k = orc.new_field(); k[:V // 2] = random_spinor(5, V // 2)
    l, ref = orc.new_field(), orc.new_field()
    d.Hopping_Matrix(0, l.ctypes.data_as(VP), k.ctypes.data_as(VP))
    orc.Hopping_Matrix(0, ref, k)
    assert rel_err(l[:V // 2], ref[:V // 2]) < TOL
    out = tmp_path / "dropin.lime"
    assert d.write_gauge_field(str(out).encode(), 32, None) == 0
    assert out.read_bytes() == open(os.path.join(GOLD, "ildg_%dx%dx%dx%d_prec32.lime" % (T, LX, LY, LZ)), "rb").read()

    # with a paramsXlfInfo (io/params.h:71-88) the first record is the plain-text message of io/utils_write_xlf.c:35-55 -- what the
    # reference's write_gauge_field writes (io/gauge_write.c:35) -- and the file is the 64-bit fixture byte for byte
    class Xlf(C.Structure):
        _fields_ = [("date", C.c_char * 64), ("package_version", C.c_char * 32), ("beta", C.c_double), ("c2_rec", C.c_double), ("epsilonbar", C.c_double),
                    ("kappa", C.c_double), ("mu", C.c_double), ("mubar", C.c_double), ("plaq", C.c_double), ("counter", C.c_int), ("time", C.c_long)]
    x = META["xlf_info"]
    xi = Xlf(date=x["date"].encode(), package_version=x["package_version"].encode(), beta=x["beta"], c2_rec=x["c2_rec"], epsilonbar=x["epsilonbar"],
             kappa=x["kappa"], mu=x["mu"], mubar=x["mubar"], plaq=x["plaq"], counter=x["counter"], time=x["time"])
    out64 = tmp_path / "dropin64.lime"
    assert d.write_gauge_field(str(out64).encode(), 64, C.byref(xi)) == 0
    assert out64.read_bytes() == open(fixture, "rb").read()
    xi.kappa = 0.0                                                     # the other branch of utils_write_xlf.c
    assert d.write_gauge_field(str(out64).encode(), 64, C.byref(xi)) == 0
    want0 = ("plaquette = %e\n trajectory nr = %d\n beta = %.12f\n kappa = %.12f\n 2*kappa*mu = %.12f\n c2_rec = %f\n date = %s"
             % (x["plaq"], x["counter"], x["beta"], 0.0, x["mu"], x["c2_rec"], x["date"])).encode()
    assert want0 in out64.read_bytes()[:1024]
    stub.stub_set_io(32, 0)
    assert d.read_gauge_field(fixture.encode(), gfpp) == -1           # GaugeConfigReadPrecision = 32 against a 64-bit file
    stub.stub_set_io(64, 0)
    d.tmlqcd_hip_finalize()
