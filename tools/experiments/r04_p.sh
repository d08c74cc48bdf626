# self-attention d = 40 / 80: K / V staged by LDS-DMA (planes) vs through registers -- parity, then same-box A/B
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=eta-inversion_amd/etainv/lib
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or attn" > gpurun_out/r04/p_kernels.log 2>&1; tail -3 gpurun_out/r04/p_kernels.log
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_realsize_gpu.py tests/test_configs_gpu.py -x -q > gpurun_out/r04/p_unet.log 2>&1; tail -3 gpurun_out/r04/p_unet.log
python tools/ab_ops.py --a $L/libetainv_hip_a40reg.so --b $L/libetainv_hip.so --only attn --rounds 3 > gpurun_out/r04/p_ab.log 2>&1
cat gpurun_out/r04/p_ab.log
