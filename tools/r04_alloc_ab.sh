# direct carrier: what the receive buffers are made of (plain hipMalloc / fine-grained / uncached), T_local 8 and 32, both forms
mkdir -p gpurun_out
RUN="timeout -k 10 90 python bench.py --no-cpu --no-rows --steps 300 --warmup 30 --cg-iters 30"
export TMLQCD_HIP_FLAG_TIMEOUT_S=5
for T in 8 32; do
  $RUN --T $T > gpurun_out/r04_aa_unsplit_T$T.json 2>/dev/null
  $RUN --T $T --loopback 1 > gpurun_out/r04_aa_lb1_T$T.json 2>/dev/null
  for k in plain finegrained uncached; do
    for f in 1 0; do
      TMLQCD_HIP_DIRECT_ALLOC=$k $RUN --T $T --loopback 3 --opt direct_form=$f > gpurun_out/r04_aa_${k}_f${f}_T$T.json 2>gpurun_out/r04_aa_${k}_f${f}_T$T.err || { echo "$k form $f T=$T failed"; exit 1; }
    done
  done
done
python - <<'PY'
import json
for T in (8,32):
    u=json.load(open('gpurun_out/r04_aa_unsplit_T%d.json'%T))
    for tag in ['lb1']+['%s_f%d'%(k,f) for k in ('plain','finegrained','uncached') for f in (1,0)]:
        d=json.load(open('gpurun_out/r04_aa_%s_T%d.json'%(tag,T)))
        print("T=%2d %-16s unsplit %.4f | split %.4f ms/step (%.1f %%) cg %.0f (%.1f %%)" % (T, tag, u['ms_per_step'], d['ms_per_step'], 100*u['ms_per_step']/d['ms_per_step'], d['cg']['iters_per_s'], 100*d['cg']['iters_per_s']/u['cg']['iters_per_s']))
PY
