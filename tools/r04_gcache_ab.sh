# T-split ranks whose gauge copy fits the Infinity Cache (4 x 32^3: 151 MB): links kept there ("gauge_cache" -1, automatic) vs streamed
# (gauge_cache=0, the split kernels' only form until now); same box, alternating.  The unsplit kernel always used the cache at this size.
mkdir -p gpurun_out
RUN="timeout -k 10 90 python bench.py --no-cpu --no-rows --steps 300 --warmup 30 --cg-iters 30"
export TMLQCD_HIP_FLAG_TIMEOUT_S=5
for T in ${TS:-4}; do
  $RUN --T $T > gpurun_out/gc_u_$T.json 2>gpurun_out/gc.err || exit 1
  $RUN --T $T --opt gauge_cache=0 > gpurun_out/gc_u0_$T.json 2>gpurun_out/gc.err || exit 1
  for rep in 1 2; do
    for v in auto 0; do
      opt=""; [ $v = 0 ] && opt="--opt gauge_cache=0"
      $RUN --T $T --loopback 3 --opt direct_form=1 $opt > gpurun_out/gc_d_${v}_${T}_$rep.json 2>gpurun_out/gc.err || exit 1
      $RUN --T $T --loopback 1 $opt > gpurun_out/gc_c_${v}_${T}_$rep.json 2>gpurun_out/gc.err || exit 1
    done
  done
done
python - <<'PY'
import json, os
for T in [int(t) for t in os.environ.get("TS", "4").split()]:
    u = json.load(open('gpurun_out/gc_u_%d.json' % T)); u0 = json.load(open('gpurun_out/gc_u0_%d.json' % T))
    print("T_local %2d unsplit: links cached %.4f ms/step cg %.0f | streamed %.4f ms/step cg %.0f" % (T, u['ms_per_step'], u['cg']['iters_per_s'], u0['ms_per_step'], u0['cg']['iters_per_s']))
    for form, tag in (("direct carrier, one kernel", "d"), ("default form (copies)", "c")):
        for v in ("auto", "0"):
            for rep in (1, 2):
                d = json.load(open('gpurun_out/gc_%s_%s_%d_%d.json' % (tag, v, T, rep)))
                print("T_local %2d %-27s gauge_cache %-4s run %d: %.4f ms/step (%.1f %% of unsplit)  cg %.0f it/s (%.1f %%)" % (T, form, v, rep, d['ms_per_step'], 100 * u['ms_per_step'] / d['ms_per_step'], d['cg']['iters_per_s'], 100 * d['cg']['iters_per_s'] / u['cg']['iters_per_s']))
PY
