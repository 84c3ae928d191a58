#!/usr/bin/env python3
"""Register / scratch / occupancy guard of the stencil kernels (round-3 review, item 3).

The build (tmlqcd_amd/csrc/Makefile) compiles every .hip file with -Rpass-analysis=kernel-resource-usage and keeps the
compiler's remarks next to the object (lib/<name>.ru.txt).  This script parses them, writes the table
(profiles/r04_resource_usage.txt) and FAILS the build when an instance the default dispatch of hopping_impl.inc can launch
spills (scratch > 0) or drops below three waves per SIMD -- the split path once lost 40 % to a spill nobody looked at
(profiles/r03_split_forms.md, last paragraph of `split_early`).

    check_resources.py [--table OUT.txt] lib/hopping.ru.txt lib/hopping32.ru.txt lib/clover.ru.txt
"""
import re
import subprocess
import sys

FIELDS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")
CXXFILT = "c++filt"

# Instances the default dispatch reaches (launch_variant / launch_exterior / launch_pack in hopping_impl.inc), as regular
# expressions on the demangled name, with the floor each must keep.  Template arguments of hop_kernel:
#   <EPI, TSKIP, NTIO, BS, MINW, GAUX, STG>
# EPI 5 / 6 / 7 / 10 / 11 are the clover epilogues (a 6x6 block product per chirality on top of the stencil, up to 256 VGPRs + a few
# AGPRs) and TSKIP 2 is the per-lane form of ragged test lattices: those are held to "no scratch"; everything else to "no scratch" and
# three waves per SIMD.
CLOVER_EPI = {5, 6, 7, 10, 11}
RULES = [
    # fp64, large lattices: the LDS-staged kernel, __launch_bounds__(256, 3) for the twisted-mass epilogues, (256, 1) for the clover ones
    (r"^void hop64::hop_kernel<(\d+), ([013]), true, 256, (3|1), -1, 64>", "fp64 staged"),
    # fp64, small lattices and ragged shapes: the gather kernel
    (r"^void hop64::hop_kernel<(\d+), ([0123]), true, (64|256), (1|3), (-1|-2), 0>", "fp64 gather"),
    (r"^void hop64::hop_split4_kernel<(\d+), true>", "fp64 hop-split"),
    (r"^void hop64::hop_exterior_kernel<(\d+), true>", "fp64 exterior"),
    (r"^void hop64::pack_faces_kernel", "fp64 pack"),
    # fp32: gather kernel by default ("lds32" 0), the staged one behind the option
    (r"^void hop32::hop_kernel<(\d+), ([0123]), true, (64|256), 1, (-1|-2), 0>", "fp32 gather"),
    (r"^void hop32::hop_exterior_kernel<(\d+), true>", "fp32 exterior"),
    (r"^void hop32::pack_faces_kernel", "fp32 pack"),
    # the two plaquette-leaf kernels of the clover rows (interior instances): 230 - 254 VGPRs by design, two waves per SIMD, no scratch
    (r"^void sw_term_kernel<SwFastLd>", "sw_term", 2),
    (r"^void sw_all_gather_kernel<SwFastLd>", "sw_all gather", 2),
]


def parse(path):
    out = []
    cur = None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1), "file": path}
            out.append(cur)
            continue
        m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1) in FIELDS:
            cur[m.group(1)] = int(m.group(2)) if m.group(2).lstrip("-").isdigit() else m.group(2)
    return out


def demangle(names):
    try:
        r = subprocess.run([CXXFILT], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True)
        return r.stdout.split("\n")[: len(names)]
    except (OSError, subprocess.CalledProcessError):
        return names


def main(argv):
    table = None
    files = []
    it = iter(argv)
    for a in it:
        if a == "--table":
            table = next(it)
        else:
            files.append(a)
    if not files:
        print(__doc__)
        return 2
    ks = []
    for f in files:
        ks += parse(f)
    for k, n in zip(ks, demangle([k["mangled"] for k in ks])):
        k["name"] = re.sub(r"\(.*$", "", n)   # drop the parameter list
    bad, rows, guarded = [], [], 0
    for k in sorted(ks, key=lambda k: k["name"]):
        tag, floor = "", None
        for rule in RULES:
            rx, what = rule[0], rule[1]
            m = re.match(rx, k["name"])
            if m and len(rule) > 2:
                floor = rule[2]
                tag = "%s (>= %d waves)" % (what, floor)
                break
            if m:
                g = m.groups()
                epi = int(g[0]) if g else -1
                ragged = len(g) > 1 and g[1] == "2"
                floor = 1 if (epi in CLOVER_EPI or ragged) else 3
                # the one-kernel form of the direct carrier sizes its wait budget on two waves per SIMD for the clover epilogues (direct_one_kernel)
                if epi in CLOVER_EPI and len(g) > 1 and g[1] == "3":
                    floor = 2
                tag = "%s (>= %d waves)" % (what, floor)
                break
        scratch, occ = k.get("ScratchSize [bytes/lane]", -1), k.get("Occupancy [waves/SIMD]", -1)
        verdict = ""
        if floor is not None:
            guarded += 1
            if scratch != 0 or occ < floor:
                verdict = "FAIL"
                bad.append(k)
            else:
                verdict = "ok"
        rows.append("%-96s %5s %5s %7s %4s %7s  %-28s %s" % (k["name"][:96], k.get("VGPRs", "?"), k.get("AGPRs", "?"), scratch, occ,
                                                          k.get("LDS Size [bytes/block]", "?"), tag, verdict))
    head = "%-96s %5s %5s %7s %4s %7s  %-28s %s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS", "guard", "")
    text = "\n".join([head, "-" * len(head)] + rows) + "\n"
    text += "\n%d kernels, %d guarded, %d failing\n" % (len(ks), guarded, len(bad))
    if table:
        with open(table, "w") as f:
            f.write("# -Rpass-analysis=kernel-resource-usage of the stencil translation units, gfx950 (tools/check_resources.py;\n"
                    "# written by every build: tmlqcd_amd/csrc/Makefile).  guard = an instance the default dispatch can launch.\n")
            f.write(text)
    if guarded == 0:
        print("check_resources: no guarded kernel found in", files, file=sys.stderr)
        return 1
    if bad:
        print(text)
        print("check_resources: %d default-dispatch kernel(s) spill or run below their occupancy floor:" % len(bad), file=sys.stderr)
        for k in bad:
            print("   %s: scratch %s B/lane, %s waves/SIMD, %s VGPRs" % (k["name"], k.get("ScratchSize [bytes/lane]"), k.get("Occupancy [waves/SIMD]"), k.get("VGPRs")), file=sys.stderr)
        return 1
    print("check_resources: %d kernels, %d guarded, all within budget" % (len(ks), guarded))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
