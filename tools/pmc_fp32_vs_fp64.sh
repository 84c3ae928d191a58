#!/bin/bash
# counters of the plain fp64 and fp32 stencils side by side (one rocprofv3 --pmc pass per group, counters only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcg_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcg_$i -- python3 $R/tools/pmc_fp32_vs_fp64.py > $OUT/pmcg_$i.log 2>&1 || { echo "group '$grp' refused"; continue; }
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/pmcg_$i/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "hop_kernel<0, 0" in n:
            acc[("fp64" if "hop64" in n else "fp32", r["Counter_Name"])].append(float(r["Counter_Value"]))
for (p,c),v in sorted(acc.items(), key=lambda kv:(kv[0][1],kv[0][0])):
    print("%-34s %s  %14.0f  per site %10.3f  (n=%d)" % (c, p, sum(v)/len(v), sum(v)/len(v)/524288, len(v)))
PY
done
