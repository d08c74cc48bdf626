mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_e2e_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -s > gpurun_out/r03/t_skip.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_skip.log
python tools/parity_s50.py --subjects hip_fp16 hip_bf16 hip_fp32 ref_fp16 ref_bf16 --ref-pairs 1 --out gpurun_out/r03/parity_S50_final.json > gpurun_out/r03/parity_final.log 2>&1; echo "rc $?" >> gpurun_out/r03/parity_final.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_skip.json 2> gpurun_out/r03/bench_b32_skip.err
ETAINV_NO_DEAD_ROW_SKIP=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_noskip.json 2> gpurun_out/r03/bench_b32_noskip.err
python bench.py --config 2 --steps 3 --warmup 1 > gpurun_out/r03/bench_cfg2_skip.json 2> gpurun_out/r03/bench_cfg2_skip.err
timeout 900 python -m pytest tests/test_properties_gpu.py tests/test_batch_gpu.py tests/test_modules_api_gpu.py tests/test_snapshot_gpu.py -q -m gpu -x > gpurun_out/r03/t_more.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_more.log
grep -h -E "passed|failed|^rc|dead-row" gpurun_out/r03/t_skip.log gpurun_out/r03/t_more.log gpurun_out/r03/parity_final.log
