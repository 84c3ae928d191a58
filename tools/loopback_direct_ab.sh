# T_local = 8 / 16 / 32 slabs of 32^3: unsplit, split with self-exchange by copies (loopback 1), and with the pack kernel writing
# straight into the receive buffers (loopback 3: what faces written into an IPC-mapped neighbour buffer would save)
for T in 8 16 32; do
  for lb in 0 1 3; do
    if [ $lb = 0 ]; then extra=""; else extra="--loopback $lb"; fi
    python bench.py --T $T $extra --no-cpu --steps 300 --warmup 30 --cg-iters 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('T_local=%2d loopback=%s  ms/step %.4f  us/launch %.1f  cg %.0f it/s' % ($T, '$lb', d['ms_per_step'], d['roofline']['us_per_launch'], d['cg']['iters_per_s']))"
  done
done
python tools/hopsplit_ab.py 12 16 20 24 2>&1 | grep -v "hopsplit=0" | awk 'NR%2==1'
