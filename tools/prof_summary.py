"""Condense the rocprofv3 CSVs of tools/profile.sh into gpurun_out/<tag>_summary.{md,json}.

HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE (KiB) under-reports wide
coalesced reads by exactly 2x on gfx950 -> doubled (calibrated in the same run on the linalg
streams whose byte count is known: dotr_kernel reads 384 B/site and reports 192); WRITE_SIZE exact."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
Vh = int(sys.argv[2]) if len(sys.argv) > 2 else 32 ** 4 // 2


def one(pattern):
    g = glob.glob(os.path.join(out, pattern))
    return g[0] if g else None


stats = []
f = one("%s_stats/*/*_kernel_stats.csv" % tag)
if f:
    for r in csv.DictReader(open(f)):
        stats.append((r["Name"], int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]), float(r["MinNs"]), float(r["MaxNs"])))
pmc = {}
for name in ("fetch", "write"):
    f = one("%s_pmc_%s/*/*_counter_collection.csv" % (tag, name))
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    pmc[name] = {k: sum(v) / len(v) for k, v in acc.items() if len(v) >= 2}
    vg = {}
    if f:
        for r in csv.DictReader(open(f)):
            vg[r["Kernel_Name"]] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
    pmc[name + "_regs"] = vg
lines = ["# rocprofv3 summary `%s` (bench.py, 32^4 fp64, one MI355X)" % tag, "",
         "| kernel | calls | avg us | min us | max us | % | read B/site (2xFETCH) | write B/site | traffic B/launch | VGPR |",
         "|---|---|---|---|---|---|---|---|---|---|"]
js = {"tag": tag, "kernels": {}}
for name, calls, avg, pct, mn, mx in stats:
    rd = pmc["fetch"].get(name)
    wr = pmc["write"].get(name)
    rdb = rd * 1024 * 2 if rd is not None else None
    wrb = wr * 1024 if wr is not None else None
    tot = (rdb + wrb) if (rdb is not None and wrb is not None) else None
    regs = pmc["fetch_regs"].get(name, ("", "", "", ""))
    lines.append("| `%s` | %d | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s |" % (
        name[:70], calls, avg / 1e3, mn / 1e3, mx / 1e3, pct,
        "%.0f" % (rdb / Vh) if rdb is not None else "-", "%.0f" % (wrb / Vh) if wrb is not None else "-",
        "%.4g" % tot if tot is not None else "-", regs[0]))
    js["kernels"][name] = {"calls": calls, "avg_us": avg / 1e3, "read_bytes": rdb, "write_bytes": wrb, "traffic_bytes": tot}
md = "\n".join(lines) + "\n"
open(os.path.join(out, "%s_summary.md" % tag), "w").write(md)
hop = [k for k in js["kernels"] if "hop_kernel<0, 0, true, 256, 3, -1, 64>" in k]
if hop and js["kernels"][hop[0]]["traffic_bytes"]:
    js["bytes_per_launch"] = js["kernels"][hop[0]]["traffic_bytes"]
    js["kernel"] = hop[0]
    js["avg_us"] = js["kernels"][hop[0]]["avg_us"]
json.dump(js, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
print(md)
