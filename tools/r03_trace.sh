cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in comm nocom unsplit; do
rm -rf gpurun_out/tr_$m
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_$m -- python tools/split_timeline.py 32 ${TT:-8} $m > gpurun_out/tr_$m.log 2>&1
echo "== $m"
python tools/trace_print.py $(ls gpurun_out/tr_$m/*/*kernel_trace.csv | head -1) 14
done
