mkdir -p gpurun_out
TMLQCD_HIP_FLAG_TIMEOUT_S=20 timeout -k 10 200 python -m pytest tests/test_gpu_operators.py -x -q -m gpu -k "one_communicator" > gpurun_out/r03_t3.log 2>&1 || { tail -40 gpurun_out/r03_t3.log; exit 1; }
tail -2 gpurun_out/r03_t3.log
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err || { echo "bench failed"; tail -20 gpurun_out/r03_bench_n1.err; }
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_n1.json')); print({k: d[k] for k in ('value','ms_per_step','roofline','cg_16','nocom')}); print(d['cg']['iters_per_s'], d['cpu_baseline']['value'], d.get('parity_max_rel_err_vs_cpu'), d['next_rows'])"
TMLQCD_HIP_FLAG_TIMEOUT_S=20 timeout -k 10 500 python bench.py --loopback 2 --rehearse-split --no-cpu --no-rows --steps 100 --warmup 10 --cg-iters 25 > gpurun_out/r03_bench_rehearse.json 2> gpurun_out/r03_bench_rehearse.err || { echo "rehearse failed"; tail -20 gpurun_out/r03_bench_rehearse.err; }
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_rehearse.json')); print(json.dumps({k: d.get(k) for k in ('rank_check','strong','strong_32','comm_split','rccl_nranks')}, indent=1))"
