mkdir -p gpurun_out/r03
python tools/parity_s50.py --subjects hip_fp16 hip_bf16 hip_fp32 --out gpurun_out/r03/parity_S50_hip.json > gpurun_out/r03/parity_hip.log 2>&1; echo "rc $?" >> gpurun_out/r03/parity_hip.log
timeout 1200 python -m pytest tests/test_fp32_gpu.py -q -m gpu -s > gpurun_out/r03/t_fp32_b.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_fp32_b.log
timeout 1500 python -m pytest tests/test_configs_gpu.py -q -m gpu -s > gpurun_out/r03/t_configs.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_configs.log
timeout 1500 python -m pytest tests/test_realsize_gpu.py -q -m gpu -s > gpurun_out/r03/t_realsize.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_realsize.log
lscpu > gpurun_out/r03/lscpu.txt
grep -h -E "passed|failed|rc " gpurun_out/r03/parity_hip.log gpurun_out/r03/t_fp32_b.log gpurun_out/r03/t_configs.log gpurun_out/r03/t_realsize.log
