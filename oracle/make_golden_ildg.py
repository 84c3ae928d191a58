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
# the "xlf-info" message exactly as write_gauge_field writes it (io/gauge_write.c:35 -> io/utils_write_xlf.c:35-55: plain text), for a
# paramsXlfInfo as construct_paramsXlfInfo fills it (io/params_construct_xlfInfo.c: date = ctime(), trailing newline included)
XLF = dict(plaq=0.5, counter=7, beta=3.9, kappa=0.160856, mu=0.004, c2_rec=0.0, time=1234567890, package_version="5.2.0", mubar=0.0,
           epsilonbar=0.0, date="Fri Feb 13 23:31:30 2009\n")


def xlf_info_text(plaq, counter, beta, kappa, mu, c2_rec, time, package_version, mubar, epsilonbar, date):
    if kappa != 0.0:
        return ("plaquette = %14.12f\n trajectory nr = %d\n beta = %.12f, kappa = %.12f, mu = %.12f, c2_rec = %f\n time = %d\n"
                " hmcversion = %s\n mubar = %.12f\n epsilonbar = %.12f\n date = %s"
                % (plaq, counter, beta, kappa, mu, c2_rec, time, package_version, mubar, epsilonbar, date))
    return ("plaquette = %e\n trajectory nr = %d\n beta = %.12f\n kappa = %.12f\n 2*kappa*mu = %.12f\n c2_rec = %f\n date = %s"
            % (plaq, counter, beta, kappa, mu, c2_rec, date))


g = syn.gauge_field(41, T, LX, LY, LZ)
out = {"lattice": [T, LX, LY, LZ], "gauge_seed": 41, "files": {}, "crc32_known_answers": [], "xlf_info": XLF, "xlf_info_text": xlf_info_text(**XLF)}
assert ib.ref_available(), "build oracle/_ref first (make -C oracle ref)"
for prec in (64, 32):
    name = "ildg_%dx%dx%dx%d_prec%d.lime" % (T, LX, LY, LZ, prec)
    path = os.path.join(ROOT, "tests", "golden", name)
    xlf = xlf_info_text(**XLF) if prec == 64 else None
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
