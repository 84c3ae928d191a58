mkdir -p gpurun_out
TMLQCD_HIP_FLAG_TIMEOUT_S=20 timeout -k 10 900 python -m pytest tests -x -q -m gpu --timeout 400 --timeout-method=thread > gpurun_out/r03_gpu_all.log 2>&1
tail -5 gpurun_out/r03_gpu_all.log
bash tools/profile.sh r03 > gpurun_out/r03_profile.log 2>&1; tail -30 gpurun_out/r03_profile.log
bash tools/profile_rows.sh r03 > gpurun_out/r03_profile_rows.log 2>&1; tail -25 gpurun_out/r03_profile_rows.log
