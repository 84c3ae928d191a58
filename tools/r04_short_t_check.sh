# after the short-lattice rules (slab order below 24 time-slices, 256-thread blocks from 131072 sites per parity): unsplit and direct-carrier
# rates at T_local 4 / 8 / 16 (32^3) and 12 x 48^3, against the forced old choice (--opt xcd=4 = tile order; T = 8 also block 64)
mkdir -p gpurun_out
RUN="timeout -k 10 120 python bench.py --no-cpu --no-rows --steps 300 --warmup 30 --cg-iters 30"
export TMLQCD_HIP_FLAG_TIMEOUT_S=5
one() {  # tag, args...
  tag=$1; shift
  $RUN "$@" > gpurun_out/st.json 2>gpurun_out/st.err || { echo "$tag: failed"; return; }
  python -c "
import json; d=json.load(open('gpurun_out/st.json')); print('%-58s %.4f ms/step  cg %.0f it/s' % ('$tag', d['ms_per_step'], d['cg']['iters_per_s']))"
}
for rep in 1 2; do
  one " 8 x 32^3 unsplit, new rules"            --T 8
  one " 8 x 32^3 unsplit, old (block 64, tile)" --T 8 --opt block=64 --opt xcd=4
  one " 8 x 32^3 direct,  new rules"            --T 8 --loopback 3
  one " 8 x 32^3 direct,  old (block 64, tile)" --T 8 --loopback 3 --opt block=64 --opt xcd=4
  one "16 x 32^3 unsplit, new rules"            --T 16
  one "16 x 32^3 unsplit, old (tile)"           --T 16 --opt xcd=4
  one "16 x 32^3 direct,  new rules"            --T 16 --loopback 3
  one "16 x 32^3 direct,  old (tile)"           --T 16 --loopback 3 --opt xcd=4
  one " 4 x 32^3 unsplit"                       --T 4
  one " 4 x 32^3 direct"                        --T 4 --loopback 3
done
