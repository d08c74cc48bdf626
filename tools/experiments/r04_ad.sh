mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv3x3" > gpurun_out/r04/ad_conv.log 2>&1; tail -2 gpurun_out/r04/ad_conv.log
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_realsize_gpu.py tests/test_configs_gpu.py tests/test_properties_gpu.py tests/test_s50_gpu.py tests/test_e2e_gpu.py -x -q > gpurun_out/r04/ad_unet.log 2>&1; tail -2 gpurun_out/r04/ad_unet.log
python tools/unet_call.py --rows 128 --calls 3 --shapes > gpurun_out/r04/ad_shapes.log 2>&1
grep -E "== igemm|total event|9.664e\+11" gpurun_out/r04/ad_shapes.log
python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', round(d['value'],4), round(d['ms_per_step'],1))"
