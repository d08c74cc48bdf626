mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_fp32_gpu.py -q -m gpu -s > gpurun_out/r03/t_unet_fp32_c.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_unet_fp32_c.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_share.json 2> gpurun_out/r03/bench_b32_share.err
ETAINV_NO_PREFIX_SHARE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_noshare.json 2> gpurun_out/r03/bench_b32_noshare.err
python bench.py --config 2 --steps 3 --warmup 1 > gpurun_out/r03/bench_cfg2.json 2> gpurun_out/r03/bench_cfg2.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_torchrun1.json 2> gpurun_out/r03/bench_torchrun1.err
python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_gpus2_on_1gpu.out 2> gpurun_out/r03/bench_gpus2_on_1gpu.err; echo "rc $?" >> gpurun_out/r03/bench_gpus2_on_1gpu.out
timeout 1500 python -m pytest tests/test_realsize_gpu.py tests/test_e2e_gpu.py -q -m gpu -s > gpurun_out/r03/t_realsize_e2e_c.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_realsize_e2e_c.log
grep -h -E "passed|failed|^rc" gpurun_out/r03/t_unet_fp32_c.log gpurun_out/r03/t_realsize_e2e_c.log gpurun_out/r03/bench_gpus2_on_1gpu.out
