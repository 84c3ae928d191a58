# round-3 closing run on one MI355X: the whole GPU suite, the default bench line, the multi-rank legs rehearsed, the three rocprofv3 passes, the rows
mkdir -p gpurun_out
TMLQCD_HIP_FLAG_TIMEOUT_S=20 timeout -k 10 900 python -m pytest tests -x -q -m gpu --timeout 400 --timeout-method=thread > gpurun_out/r03_gpu_all.log 2>&1
tail -4 gpurun_out/r03_gpu_all.log
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err || { echo "bench failed"; tail -20 gpurun_out/r03_bench_n1.err; }
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_n1.json')); print({k: d[k] for k in ('value','ms_per_step','cg_16')}, d['roofline']['frac'], d['cg']['iters_per_s'], d['cpu_baseline']['value'], d.get('parity_max_rel_err_vs_cpu'), d['next_rows'])"
TMLQCD_HIP_FLAG_TIMEOUT_S=20 timeout -k 10 500 python bench.py --loopback 2 --rehearse-split --no-cpu --no-rows --steps 100 --warmup 10 --cg-iters 25 > gpurun_out/r03_bench_rehearse.json 2> gpurun_out/r03_bench_rehearse.err || { echo "rehearse failed"; tail -20 gpurun_out/r03_bench_rehearse.err; }
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_rehearse.json')); print(d['rank_check']['ok'], d['strong']['value'], d['strong_32']['value'], d['strong_32']['rank_check']['ok'])"
bash tools/profile.sh r03 > gpurun_out/r03_profile.log 2>&1; head -12 gpurun_out/r03_summary.md
bash tools/profile_rows.sh r03 > gpurun_out/r03_profile_rows.log 2>&1; tail -16 gpurun_out/r03_profile_rows.log
