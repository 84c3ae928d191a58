# kernel timelines of the split forms at T_local = $TT (default 8): unsplit, default form (loopback 1), direct carrier one-kernel / two-kernel (loopback 3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TMLQCD_HIP_FLAG_TIMEOUT_S=5
run() {   # tag mode loopback opts
  rm -rf gpurun_out/tr4_$1
  TL_OPTS=$4 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr4_$1 -- python tools/split_timeline.py 32 ${TT:-8} $2 $3 > gpurun_out/tr4_$1.log 2>&1
  echo "== $1"
  python tools/trace_print.py $(ls gpurun_out/tr4_$1/*/*kernel_trace.csv | head -1) ${ROWS:-12}
}
run unsplit unsplit 0 ""
run default comm 1 ""
run direct_one comm 3 direct_form=1
run direct_two comm 3 direct_form=0
