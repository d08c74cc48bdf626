set -x
mkdir -p gpurun_out/r03
python -m pytest tests/test_unet_gpu.py tests/test_modules_api_gpu.py -x -q -m gpu -k "cache or dpm" > gpurun_out/r03/t_new.log 2>&1; echo "rc new $?" >> gpurun_out/r03/t_new.log
(python tools/parity_s50.py --subjects hip_fp16 hip_bf16 ref_fp16 ref_bf16 --cache-out gpurun_out/parity_cache --out gpurun_out/r03/parity_S50_a.json > gpurun_out/r03/parity_a.log 2>&1; echo "rc $?" >> gpurun_out/r03/parity_a.log) &
PAR=$!
python -m pytest tests/test_e2e_gpu.py -q -m gpu -s > gpurun_out/r03/t_e2e.log 2>&1; echo "rc e2e $?" >> gpurun_out/r03/t_e2e.log
wait $PAR
python -m pytest tests/test_realsize_gpu.py -q -m gpu -s > gpurun_out/r03/t_realsize.log 2>&1; echo "rc realsize $?" >> gpurun_out/r03/t_realsize.log
lscpu > gpurun_out/r03/lscpu.txt
tail -3 gpurun_out/r03/*.log
