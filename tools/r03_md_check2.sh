mkdir -p gpurun_out
TMLQCD_HIP_FLAG_TIMEOUT_S=20 timeout -k 10 900 python -m pytest tests/test_gpu_force.py tests/test_gpu_md_trajectory.py tests/test_gpu_hopping.py tests/test_gpu_clover.py -x -q -m gpu > gpurun_out/r03_t5.log 2>&1 || { tail -40 gpurun_out/r03_t5.log; exit 1; }
tail -3 gpurun_out/r03_t5.log
timeout -k 10 200 python tools/next_rows_speed.py > gpurun_out/r03_next_rows_speed.log 2>&1; grep -n "update_" gpurun_out/r03_next_rows_speed.log
