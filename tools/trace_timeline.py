"""Print a compact timeline (kernel, queue, start offset, duration) from a rocprofv3 kernel-trace CSV."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0][-48:]
# locate the steady-state region: last N rows
sel = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-50s q%-3s start %9.1f us  dur %7.1f us  grid %s" % (name(r), r["Queue_Id"], s / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", "")))
