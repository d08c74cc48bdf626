mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_unet_gpu.py tests/test_realsize_gpu.py tests/test_configs_gpu.py tests/test_s50_gpu.py -x -q > gpurun_out/r04/aa_tests.log 2>&1; tail -4 gpurun_out/r04/aa_tests.log
for i in 1 2; do
python tools/unet_call.py --rows 128 --calls 3 --shapes > gpurun_out/r04/aa_shapes_dma_$i.log 2>&1
ETAINV_A40_DMA=0 python tools/unet_call.py --rows 128 --calls 3 --shapes > gpurun_out/r04/aa_shapes_reg_$i.log 2>&1
done
for f in gpurun_out/r04/aa_shapes_*.log; do echo $f; grep -E "== self-attn|total event|2.749e\+12" $f; done
