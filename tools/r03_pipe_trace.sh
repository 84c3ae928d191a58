# kernel timeline of the split_pipe form (T_local = ${TT:-8}, loopback ${LB:-1}) next to the default split form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in 1 0; do
rm -rf gpurun_out/tr_pipe$p
TL_OPTS=split_pipe=$p rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_pipe$p -- python tools/split_timeline.py 32 ${TT:-8} comm ${LB:-1} > gpurun_out/tr_pipe$p.log 2>&1
echo "== split_pipe $p"
python tools/trace_print.py $(ls gpurun_out/tr_pipe$p/*/*kernel_trace.csv | head -1) ${NROWS:-26}
done
