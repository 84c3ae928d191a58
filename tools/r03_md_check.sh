mkdir -p gpurun_out
TMLQCD_HIP_FLAG_TIMEOUT_S=20 timeout -k 10 900 python -m pytest tests/test_gpu_force.py tests/test_gpu_md_trajectory.py tests/test_gpu_ildg.py tests/test_gpu_clover.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r03_t4.log 2>&1 || { tail -40 gpurun_out/r03_t4.log; exit 1; }
tail -3 gpurun_out/r03_t4.log
timeout -k 10 300 python bench.py --no-cpu --steps 100 --warmup 10 --cg-iters 25 > gpurun_out/r03_bench_rows.json 2> gpurun_out/r03_bench_rows.err || { echo "bench failed"; tail -20 gpurun_out/r03_bench_rows.err; }
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_rows.json')); print(d['next_rows'])"
timeout -k 10 200 python tools/next_rows_speed.py > gpurun_out/r03_next_rows_speed.log 2>&1; tail -25 gpurun_out/r03_next_rows_speed.log
