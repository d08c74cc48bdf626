mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_properties_gpu.py tests/test_batch_gpu.py -q -m gpu -x > gpurun_out/r03/t_ring192.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_ring192.log
tail -3 gpurun_out/r03/t_ring192.log
