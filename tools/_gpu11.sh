mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_e2e_gpu.py -q -m gpu -s > gpurun_out/r03/t_exit12.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_exit12.log
python tools/parity_s50.py --subjects hip_fp16 hip_bf16 hip_fp32 ref_fp16 ref_bf16 --ref-pairs 1 --out gpurun_out/r03/parity_S50_final3.json > gpurun_out/r03/parity_final3.log 2>&1; echo "rc $?" >> gpurun_out/r03/parity_final3.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_exit12.json 2> gpurun_out/r03/bench_b32_exit12.err
ETAINV_SRC_EXIT_9_ONLY=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_exit9only.json 2> gpurun_out/r03/bench_b32_exit9only.err
grep -h -E "passed|failed|^rc|exit after|FAILED|^hip_|^ref_" gpurun_out/r03/t_exit12.log gpurun_out/r03/parity_final3.log | cut -c1-220
