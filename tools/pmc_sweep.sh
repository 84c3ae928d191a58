#!/bin/bash
# FETCH_SIZE of the plain stencil for several mapping options (one rocprofv3 --pmc run each)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for opts in "ser=0" "ser=1" "ser=2" "ser=4"; do
  i=$((i+1))
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/sweep_$i -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu --cg-iters 1 --opt $opts > $OUT/sweep_$i.log 2>&1
  python3 - <<PY
import csv,glob
v=[]
for f in glob.glob("$OUT/sweep_$i/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void hop_kernel<0, 0") or r["Kernel_Name"].startswith("void hop64::hop_kernel<0, 0"):
            v.append(float(r["Counter_Value"]))
print("%-28s read B/site (2xFETCH) = %.0f  (n=%d)" % ("$opts", sum(v)/max(len(v),1)*2048/524288, len(v)))
PY
done
