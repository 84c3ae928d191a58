# split-path check of round 4: unsplit vs loopback 1 (copies) / 2 (one-rank RCCL) / 3 (direct carrier: one-kernel and two-kernel form) at
# T_local = 4 .. 32 on one box (32^3 spatial, fp64); % of unsplit = same-box A/B
mkdir -p gpurun_out
RUN="timeout -k 10 90 python bench.py --no-cpu --no-rows --steps 300 --warmup 30 --cg-iters 30"
export TMLQCD_HIP_FLAG_TIMEOUT_S=5
for T in 4 8 16 32; do
  $RUN --T $T > gpurun_out/r04_unsplit_T$T.json 2>gpurun_out/r04_unsplit_T$T.err || { echo "unsplit T=$T failed"; exit 1; }
  for lb in 1 2; do
    $RUN --T $T --loopback $lb > gpurun_out/r04_lb${lb}_T$T.json 2>gpurun_out/r04_lb${lb}_T$T.err || { echo "lb $lb T=$T failed"; exit 1; }
  done
  for ord in 0 1 2 3; do
    $RUN --T $T --loopback 3 --opt direct_form=1 --opt direct_order=$ord > gpurun_out/r04_lb3o${ord}_T$T.json 2>gpurun_out/r04_lb3o${ord}_T$T.err || { echo "lb 3 order $ord T=$T failed"; exit 1; }
  done
  $RUN --T $T --loopback 3 --opt direct_form=0 > gpurun_out/r04_lb3two_T$T.json 2>gpurun_out/r04_lb3two_T$T.err || { echo "lb 3 two-kernel T=$T failed"; exit 1; }
done
python - <<'PY'
import json
for T in (4,8,16,32):
    u=json.load(open('gpurun_out/r04_unsplit_T%d.json'%T))
    for tag in ('lb1','lb2','lb3o0','lb3o1','lb3o2','lb3o3','lb3two'):
        d=json.load(open('gpurun_out/r04_%s_T%d.json'%(tag,T)))
        print("T=%2d %-7s unsplit %.4f ms/step cg %.0f | split %.4f ms/step (%.1f %%) cg %.0f (%.1f %%) nocom %.4f" % (T, tag, u['ms_per_step'], u['cg']['iters_per_s'], d['ms_per_step'], 100*u['ms_per_step']/d['ms_per_step'], d['cg']['iters_per_s'], 100*d['cg']['iters_per_s']/u['cg']['iters_per_s'], d['nocom']['ms_per_step']))
PY
grep -c "gave up" gpurun_out/r04_lb*.err || true
