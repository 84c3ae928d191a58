"""CPU: the ILDG restatement (oracle/ildg_oracle.c) against what pins it -- the reference's io/dml.c + io/DML_crc32.c compiled in
place (when oracle/_ref was built), zlib's CRC-32, and the committed fixtures of tests/golden/ (checksums computed by the
reference's own checksum objects, oracle/make_golden_ildg.py).  The LIME container has no reference implementation in the tree
(c-lime is a third-party library that is not installed): its framing is checked structurally only -- parity unpinned."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import ildgbind as ib
from tmlqcd_amd import synthetic as syn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(GOLD, "ildg_checksums.json")))
T, LX, LY, LZ = META["lattice"]


def records(path):
    """(type, MB, ME, data) of every LIME record, read with nothing but the published header layout."""
    raw = open(path, "rb").read()
    pos, out = 0, []
    while pos < len(raw):
        magic, version, flags = struct.unpack(">IHB", raw[pos:pos + 7])
        assert magic == 0x456789AB and version == 1
        n = struct.unpack(">Q", raw[pos + 8:pos + 16])[0]
        typ = raw[pos + 16:pos + 144].split(b"\0")[0].decode()
        out.append((typ, bool(flags & 0x80), bool(flags & 0x40), raw[pos + 144:pos + 144 + n]))
        pos += 144 + (n + 7) // 8 * 8
    assert pos == len(raw)
    return out


def test_crc32_is_zlibs_and_the_references():
    rng = np.random.default_rng(3)
    for n in (0, 1, 3, 8, 287, 288, 576, 4097):
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert ib.crc32(b) == zlib.crc32(b) & 0xFFFFFFFF
        if ib.ref_available():
            assert ib.crc32(b) == ib.ref_crc32(b)
    for ka in META["crc32_known_answers"]:           # known answers produced by the reference's DML_crc32
        assert "%08x" % ib.crc32(bytes.fromhex(ka["hex"])) == ka["crc32"]


@pytest.mark.skipif(not ib.ref_available(), reason="oracle/_ref not built (no /root/reference here)")
def test_checksum_accumulation_matches_io_dml_c():
    rng = np.random.default_rng(4)
    rec = rng.integers(0, 256, 576 * 70, dtype=np.uint8)
    for rank0 in (0, 29 * 31 - 3, 123456):           # ranks with rank % 29 == 0 and rank % 31 == 0 are inside every window
        want = ib.ref_checksum(rec, 576, rank0)
        got = (0, 0)
        for f in range(70):
            got = ib.checksum_accum(got, rank0 + f, rec[f * 576:(f + 1) * 576].tobytes())
        assert got == want


@pytest.mark.parametrize("prec", [64, 32])
def test_fixture_files(prec):
    name = "ildg_%dx%dx%dx%d_prec%d.lime" % (T, LX, LY, LZ, prec)
    path, meta = os.path.join(GOLD, name), META["files"][name]
    assert os.path.getsize(path) == meta["bytes"]
    recs = records(path)
    names = [r[0] for r in recs]
    assert names[-3:] == ["ildg-format", "ildg-binary-data", "scidac-checksum"]                 # io/gauge_write.c:36-45
    assert [(r[1], r[2]) for r in recs[-3:]] == [(True, False), (False, False), (False, True)]   # MB / ME bits
    if prec == 64:
        assert recs[0][:3] == ("xlf-info", True, True)
    rc, gf, sums = ib.read_gauge_field(path, prec, T, LX, LY, LZ)
    assert rc == 0
    assert "%08x" % sums[0] == meta["suma"] and "%08x" % sums[1] == meta["sumb"]                 # calculated == the reference's
    assert sums[2:] == sums[:2]                                                                  # == stored in the file
    g = syn.gauge_field(META["gauge_seed"], T, LX, LY, LZ)
    if prec == 64:
        assert np.array_equal(gf, g)
    else:
        assert np.array_equal(gf, g.astype(np.float32).astype(np.float64))
    # byte order and site / link order, checked without the restatement: site f = ((t LZ + z) LY + y) LX + x, links x, y, z, t
    data = recs[-2][3]
    dt = ">f8" if prec == 64 else ">f4"
    arr = np.frombuffer(data, dtype=dt).reshape(T, LZ, LY, LX, 4, 3, 3, 2).astype(np.float64)
    want = (g if prec == 64 else g.astype(np.float32).astype(np.float64)).reshape(T, LX, LY, LZ, 4, 3, 3, 2)
    assert np.array_equal(arr.transpose(0, 3, 2, 1, 4, 5, 6, 7)[..., [3, 0, 1, 2], :, :, :], want)


def test_reader_rejects_what_the_reference_rejects(tmp_path):
    src = os.path.join(GOLD, "ildg_%dx%dx%dx%d_prec64.lime" % (T, LX, LY, LZ))
    raw = bytearray(open(src, "rb").read())
    recs = records(src)
    off = bytes(raw).index(recs[-2][3][:64])
    bad = bytearray(raw); bad[off + 1000] ^= 0x10                      # one flipped bit in the binary record: checksum mismatch
    p = tmp_path / "flipped.lime"; p.write_bytes(bad)
    assert ib.read_gauge_field(p, 64, T, LX, LY, LZ)[0] == -1
    assert ib.read_gauge_field(src, 64, T, LX, LY, LZ * 2)[0] == -1    # other lattice: record length mismatch (gauge_read_binary.c:148)
    assert ib.read_gauge_field(src, 32, T, LX, LY, LZ)[0] == -1        # other precision
    cut = raw[:len(raw) - 144 - (len(recs[-1][3]) + 7) // 8 * 8]       # no scidac-checksum record (gauge_read.c:137-142)
    p2 = tmp_path / "nochecksum.lime"; p2.write_bytes(cut)
    assert ib.read_gauge_field(p2, 64, T, LX, LY, LZ)[0] == -1


def test_round_trip_and_split_checksums(tmp_path):
    Tg, L = 4, 4
    g = syn.gauge_field(9, Tg, L, L, L)
    for prec in (64, 32):
        rec, sums = ib.pack(g, prec, Tg, L, L, L)
        back, sums2 = ib.unpack(rec, prec, Tg, L, L, L)
        assert sums == sums2
        assert np.array_equal(back, g if prec == 64 else g.astype(np.float32).astype(np.float64))
        # T-split ranks: the record is the concatenation of the ranks' parts, the checksum the XOR of theirs (io/dml.c:63-66)
        sb = 576 if prec == 64 else 288
        V2 = Tg // 2 * L ** 3
        a, sa = ib.pack(g[:V2], prec, Tg // 2, L, L, L, 0)
        b, sb2 = ib.pack(g[V2:], prec, Tg // 2, L, L, L, V2)
        assert np.array_equal(np.concatenate([a, b]), rec) and (sa[0] ^ sb2[0], sa[1] ^ sb2[1]) == sums and a.size == V2 * sb
