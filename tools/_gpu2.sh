mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_fp32_gpu.py -q -m gpu -s -k "not L64 and not 64-3 and not 64-2" > gpurun_out/r03/t_fp32_a.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_fp32_a.log
tail -40 gpurun_out/r03/t_fp32_a.log
