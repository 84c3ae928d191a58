"""TEST INFRASTRUCTURE ONLY.  Writes the ILDG fixtures of tests/golden/:
    ildg_2x4x2x6_prec64.lime, ildg_2x4x2x6_prec32.lime   (T x LX x LY x LZ = 2 x 4 x 2 x 6, links = tmlqcd_amd.synthetic.gauge_field(41, ...))
    ildg_checksums.json                                   the SciDAC checksums of their binary records, computed by the REFERENCE's
                                                          io/dml.c + io/DML_crc32.c compiled in place (oracle/_ref/libtmref_dml.so),
                                                          plus zlib.crc32 / DML_crc32 known answers on fixed byte strings
The .lime files themselves are written by the restatement (oracle/ildg_oracle.c): the reference's writer needs c-lime, which is
neither part of /root/reference nor installed -- the container framing is unpinned, the checksums and the byte strings are pinned.
Run from the repository root in the build container (needs /root/reference for `make -C oracle ref`)."""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ildgbind as ib  # noqa: E402
from tmlqcd_amd import synthetic as syn  # noqa: E402

T, LX, LY, LZ = 2, 4, 2, 6
g = syn.gauge_field(41, T, LX, LY, LZ)
out = {"lattice": [T, LX, LY, LZ], "gauge_seed": 41, "files": {}, "crc32_known_answers": []}
assert ib.ref_available(), "build oracle/_ref first (make -C oracle ref)"
for prec in (64, 32):
    name = "ildg_%dx%dx%dx%d_prec%d.lime" % (T, LX, LY, LZ, prec)
    path = os.path.join(ROOT, "tests", "golden", name)
    xlf = "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<xlf-info>\n  <plaquette>0.5</plaquette>\n  <trajectory>7</trajectory>\n</xlf-info>" if prec == 64 else None
    rc, sums = ib.write_gauge_field(path, g, prec, T, LX, LY, LZ, xlf)
    assert rc == 0
    rec, _ = ib.pack(g, prec, T, LX, LY, LZ)
    ref = ib.ref_checksum(rec, 576 if prec == 64 else 288)
    assert ref == sums, (ref, sums)
    out["files"][name] = {"prec": prec, "suma": "%08x" % ref[0], "sumb": "%08x" % ref[1], "bytes": os.path.getsize(path),
                          "checksum_by": "reference io/dml.c + io/DML_crc32.c compiled in place"}
rng = np.random.default_rng(5)
for n in (0, 1, 7, 288, 576, 1000):
    b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    assert ib.ref_crc32(b) == (zlib.crc32(b) & 0xffffffff)
    out["crc32_known_answers"].append({"hex": b.hex(), "crc32": "%08x" % ib.ref_crc32(b)})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ildg_checksums.json"), "w"), indent=1)
print(json.dumps(out["files"], indent=1))
