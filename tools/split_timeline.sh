#!/bin/bash
# kernel timeline of the split-phase stencil (loopback 1) on a T x L^3 slab: tools/split_timeline.sh L T [rows]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stl
rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -- python3 $R/tools/split_timeline.py $1 $2 > /tmp/stl.log 2>&1
python3 $R/tools/trace_timeline.py /tmp/stl ${3:-24}
