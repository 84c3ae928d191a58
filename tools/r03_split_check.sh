# split-path check of round 3: the GPU tests that exercise it, then unsplit vs loopback 1 / 2 at T_local = 4 .. 32 (one box)
mkdir -p gpurun_out
true
true
for T in 4 8 16 32; do
  timeout -k 10 60 python bench.py --T $T --no-cpu --no-rows --steps 200 --warmup 20 --cg-iters 30 > gpurun_out/r03_unsplit_T$T.json 2>gpurun_out/r03_unsplit_T$T.err || echo "unsplit T=$T failed"
  for lb in 1 2; do
  TMLQCD_HIP_FLAG_TIMEOUT_S=2 timeout -k 10 60 python bench.py --T $T --loopback $lb --no-cpu --no-rows --steps 200 --warmup 20 --cg-iters 30 > gpurun_out/r03_lb${lb}_T$T.json 2>gpurun_out/r03_lb${lb}_T$T.err || { echo "lb $lb T=$T failed"; exit 1; }
  done
done
python - <<'PY'
import json
for T in (4,8,16,32):
    u=json.load(open('gpurun_out/r03_unsplit_T%d.json'%T))
    for lb in (1,2):
        d=json.load(open('gpurun_out/r03_lb%d_T%d.json'%(lb,T)))
        print("T=%2d lb=%d unsplit %.4f ms/step cg %.0f | split %.4f ms/step (%.1f %%) cg %.0f nocom %.4f" % (T, lb, u['ms_per_step'], u['cg']['iters_per_s'], d['ms_per_step'], 100*u['ms_per_step']/d['ms_per_step'], d['cg']['iters_per_s'], d['nocom']['ms_per_step']))
PY
grep -c "gave up" gpurun_out/r03_lb*.err || true
