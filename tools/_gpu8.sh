mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_e2e_gpu.py tests/test_kernels_gpu.py tests/test_fp32_gpu.py -q -m gpu -s > gpurun_out/r03/t_skip2.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_skip2.log
grep -h -E "passed|failed|^rc|dead-row|FAILED" gpurun_out/r03/t_skip2.log
