# launch-shape options on the lattices of one rank's share of the BASELINE multi-GPU configs (unsplit kernels; the split kernels take the same
# shapes): block 64 / 256 x block order 1 / 2 (automatic) / 3 / 4
mkdir -p gpurun_out
for TL in "8 32" "12 48" "4 32" "16 32"; do
  set -- $TL; T=$1; L=$2
  for blk in 0 64 256; do
    for xcd in 2 1 3 4; do
      [ $blk = 0 ] && [ $xcd != 2 ] && continue
      timeout -k 10 120 python bench.py --T $T --L $L --no-cpu --no-rows --steps 200 --warmup 20 --cg-iters 20 --opt block=$blk --opt xcd=$xcd > gpurun_out/sw.json 2>gpurun_out/sw.err || { echo "T $T L $L block $blk xcd $xcd: failed"; continue; }
      python -c "
import json; d=json.load(open('gpurun_out/sw.json')); print('%2d x %d^3 block %3d xcd %d: %.4f ms/step  cg %.0f it/s' % ($T, $L, $blk, $xcd, d['ms_per_step'], d['cg']['iters_per_s']))"
    done
  done
done
