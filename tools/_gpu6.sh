mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/unet_call.py --rows 128 --calls 2 --shapes --dump gpurun_out/r03/launches_rows128_xcd.json > gpurun_out/r03/unet_shapes_rows128_xcd.log 2>&1
ETAINV_XCD_GN=1 python tools/unet_call.py --rows 128 --calls 2 --shapes > gpurun_out/r03/unet_shapes_rows128_xcd1.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_xcd.json 2> gpurun_out/r03/bench_b32_xcd.err
ETAINV_XCD_GN=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_xcd1.json 2> gpurun_out/r03/bench_b32_xcd1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03/pmc_fetch -- python3 tools/unet_call.py --rows 128 --calls 2 > gpurun_out/r03/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03/pmc_write -- python3 tools/unet_call.py --rows 128 --calls 2 > gpurun_out/r03/pmc_write.log 2>&1
F=$(find gpurun_out/r03/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r03/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W > gpurun_out/r03/pmc_traffic_rows128_xcd.json 2> gpurun_out/r03/pmc_traffic.err
python tools/pmc_per_launch.py gpurun_out/r03/launches_rows128_xcd.json $F $W > gpurun_out/r03/pmc_per_shape_rows128_xcd.json 2> gpurun_out/r03/pmc_per_shape.err
rm -rf gpurun_out/r03/pmc_fetch gpurun_out/r03/pmc_write
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -m gpu -x > gpurun_out/r03/t_kernels_xcd.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_kernels_xcd.log
tail -3 gpurun_out/r03/t_kernels_xcd.log
