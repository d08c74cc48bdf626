mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "conv3x3" > gpurun_out/r04/ab_conv.log 2>&1; grep -E "phase form|passed|failed|Error|error" gpurun_out/r04/ab_conv.log | tail -12
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_realsize_gpu.py tests/test_properties_gpu.py tests/test_s50_gpu.py -x -q > gpurun_out/r04/ab_unet.log 2>&1; tail -3 gpurun_out/r04/ab_unet.log
python tools/unet_call.py --rows 128 --calls 3 --shapes > gpurun_out/r04/ab_shapes_ups4.log 2>&1
ETAINV_UPS4=0 python tools/unet_call.py --rows 128 --calls 3 --shapes > gpurun_out/r04/ab_shapes_ups9.log 2>&1
for f in gpurun_out/r04/ab_shapes_*.log; do echo $f; grep -E "== igemm|total event|3.865e\+12|1.718e\+12|9.664e\+11  x  1|4.295e\+11  x  1 " $f; done
