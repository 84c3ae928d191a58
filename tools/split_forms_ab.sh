# split-phase stencil on a T_local x 32^3 slab with self-exchange (loopback 1): single-launch form ("fusedface" 1), two-kernel form (0), default (-1);
# the nocom column is the same step with communication switched off (benchmark.c:336-374)
for ff in -1 0 1; do for T in 16 32; do
  python bench.py --T $T --loopback 1 --opt fusedface=$ff --no-cpu --steps 300 --warmup 30 --cg-iters 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('T=%2d loopback 1 fusedface=%2s  ms/step %.4f  us/launch %.1f  cg %.0f it/s  nocom ms/step %.4f' % ($T, '$ff', d['ms_per_step'], d['roofline']['us_per_launch'], d['cg']['iters_per_s'], d['nocom']['ms_per_step']))"
done; done
