#!/bin/bash
# Counters of the kernels whose name contains <substring>, for any python script of tools/ (one rocprofv3 --pmc pass per counter
# group, counters only -- never combined with a trace domain):
#   tools/pmc_kernel.sh tools/<script>.py <kernel-name-substring> [units-per-launch]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SCRIPT=$1; SUB=$2; UNITS=${3:-1}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf /tmp/pmck_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmck_$i -- python3 $R/$SCRIPT > /tmp/pmck_$i.log 2>&1 || { echo "group '$grp' refused"; continue; }
  python3 - "$SUB" "$UNITS" /tmp/pmck_$i <<'PY'
import csv, glob, collections, sys
sub, units, d = sys.argv[1], float(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%-30s %-72s %16.0f  per unit %12.3f  (n=%d)" % (c, k, sum(v) / len(v), sum(v) / len(v) / units, len(v)))
PY
done
