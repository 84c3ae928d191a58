"""Print the kernels of a rocprofv3 --kernel-trace CSV as a timeline (us relative to the first), last N rows.  Usage: trace_print.py file.csv [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    nm = r["Kernel_Name"]
    nm = nm[:nm.find("(")] if "(" in nm else nm
    print("%9.2f %9.2f  dur %7.2f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), nm[:110]))
