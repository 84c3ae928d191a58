#!/bin/bash
# extra L2 counters for the stencil (separate passes, counters only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/pmc_x1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --cg-iters 2 > $OUT/pmc_x1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_x2 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --cg-iters 2 > $OUT/pmc_x2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $OUT/pmc_x3 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --cg-iters 2 > $OUT/pmc_x3.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("pmc_x1","pmc_x2","pmc_x3"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void hop_kernel<0, 0"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d,k,len(v),sum(v)/len(v), "per site", sum(v)/len(v)/524288)
PY
