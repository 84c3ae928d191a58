#!/bin/bash
# rocprofv3 --kernel-trace --stats of the rows next to the headline path (tools/next_rows_speed.py, tools/ildg_speed.py):
#   tools/profile_rows.sh <tag>   -> gpurun_out/<tag>_rows_stats.csv, gpurun_out/<tag>_rows.log
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rows_a /tmp/rows_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rows_a -- python3 $R/tools/next_rows_speed.py > $OUT/${TAG}_rows.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rows_b -- python3 $R/tools/ildg_speed.py >> $OUT/${TAG}_rows.log 2>&1
python3 - "$OUT/${TAG}_rows_stats.csv" <<'PY'
import csv, glob, sys
rows = []
for d in ("/tmp/rows_a", "/tmp/rows_b"):
    for f in glob.glob(d + "/*/*_kernel_stats.csv"):
        rows += list(csv.DictReader(open(f)))
keep = ("sw_", "deriv_Sb", "update_", "links_kernel", "halo_backward", "ildg_", "swpm", "clover_site")
with open(sys.argv[1], "w") as o:
    o.write("kernel,calls,avg_us,min_us,max_us\n")
    for r in rows:
        if any(k in r["Name"] for k in keep):
            o.write("\"%s\",%s,%.1f,%.1f,%.1f\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cat $OUT/${TAG}_rows_stats.csv
