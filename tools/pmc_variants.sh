#!/bin/bash
# HBM-side traffic and time of the stencil kernel variants selected by one option (separate counter passes, kernel-trace only):
#   tools/pmc_variants.sh <tag> <option> <v0,v1,..> [L] [T]   -> gpurun_out/<tag>.md
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
TAG=$1; OPT=$2; VALS=$3; L=${4:-32}; T=${5:-$L}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/tools/pmc_variants_run.py $OPT $VALS $L $T > $OUT/${TAG}_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $R/tools/pmc_variants_run.py $OPT $VALS $L $T > $OUT/${TAG}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $R/tools/pmc_variants_run.py $OPT $VALS $L $T > $OUT/${TAG}_write.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_l2 -- python3 $R/tools/pmc_variants_run.py $OPT $VALS $L $T > $OUT/${TAG}_l2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
Vh = $T * $L ** 3 // 2
def col(d, counter=None):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/${TAG}_%s/*/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if counter is None or r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
t = {}
for f in glob.glob("$OUT/${TAG}_stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        t[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
fe, wr, hit, miss = col("fetch"), col("write"), col("l2", "TCC_HIT_sum"), col("l2", "TCC_MISS_sum")
lines = ["# stencil variants by option $OPT = $VALS, ${T}x${L}^3, per launch (rocprofv3; read bytes = 2 x FETCH_SIZE: the gfx950 wide-read correction)", "",
         "| kernel | calls | avg us | read B/site | write B/site | L2 hits/site | L2 misses/site |", "|---|---|---|---|---|---|---|"]
for k in sorted(t):
    if "hop_kernel<0, 0" not in k: continue
    lines.append("| \`%s\` | %d | %.1f | %.0f | %.0f | %.2f | %.2f |" % (k.split("(")[0][:80], t[k][0], t[k][1], fe.get(k, 0) * 2048 / Vh, wr.get(k, 0) * 1024 / Vh, hit.get(k, 0) / Vh, miss.get(k, 0) / Vh))
open("$OUT/${TAG}.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
