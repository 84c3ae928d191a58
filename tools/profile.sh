#!/bin/bash
# rocprofv3 passes for one round (run on the GPU box through gpurun):
#   tools/profile.sh <tag>      -> gpurun_out/<tag>_{stats,pmc_fetch,pmc_write}/ + gpurun_out/<tag>_summary.{json,md}
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); never combined with trace domains other than kernel-trace.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu > $OUT/${TAG}_stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu --cg-iters 5 > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu --cg-iters 5 > $OUT/${TAG}_pmc_write.log 2>&1
echo "write pass done"
python3 $R/tools/prof_summary.py $TAG
