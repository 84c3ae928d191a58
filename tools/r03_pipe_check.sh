mkdir -p gpurun_out
for T in ${TLIST:-4 8 16 32}; do
  timeout -k 10 60 python bench.py --T $T --no-cpu --no-rows --steps 200 --warmup 20 --cg-iters 30 > gpurun_out/x_un.json 2>gpurun_out/x.err || echo "unsplit T=$T failed"
  for ea in 0 1; do
  for lb in 1 2; do
  TMLQCD_HIP_FLAG_TIMEOUT_S=2 timeout -k 10 60 python bench.py --T $T --loopback $lb --opt split_pipe=$ea --no-cpu --no-rows --steps 200 --warmup 20 --cg-iters 30 > gpurun_out/x_lb.json 2>gpurun_out/x_lb.err || { echo "lb $lb T=$T pipe=$ea failed"; tail -3 gpurun_out/x_lb.err; continue; }
  python - <<PY
import json
u=json.load(open('gpurun_out/x_un.json')); d=json.load(open('gpurun_out/x_lb.json'))
print("T=%2d lb=$lb pipe=$ea unsplit %.4f ms/step cg %.0f | split %.4f ms/step (%.1f %%) cg %.0f (%.1f %%) nocom %.4f" % ($T, u['ms_per_step'], u['cg']['iters_per_s'], d['ms_per_step'], 100*u['ms_per_step']/d['ms_per_step'], d['cg']['iters_per_s'], 100*d['cg']['iters_per_s']/u['cg']['iters_per_s'], d['nocom']['ms_per_step']))
PY
  echo "give-ups: $(grep -c 'gave up' gpurun_out/x_lb.err)"
  done
  done
done
true
