mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_e2e_gpu.py -q -m gpu -x > gpurun_out/r03/t_kvfix.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_kvfix.log
tail -3 gpurun_out/r03/t_kvfix.log
