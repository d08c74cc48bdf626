mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=eta-inversion_amd/etainv/lib
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_properties_gpu.py -x -q > gpurun_out/r04/u_kernels.log 2>&1; tail -2 gpurun_out/r04/u_kernels.log
ETAINV_LIB=$PWD/$L/libetainv_hip_stamps.so ETAINV_IGEMM_STAMPS=1 python tools/experiments/r04_stamps_conv.py > gpurun_out/r04/u_stamps.log 2>&1
grep -v amdgpu.ids gpurun_out/r04/u_stamps.log | grep -v "clock"
python tools/ab_ops.py --a $L/libetainv_hip_prev.so --b $L/libetainv_hip.so --rounds 3 > gpurun_out/r04/u_ab.log 2>&1
grep -E "conv|lin|geglu|shape" gpurun_out/r04/u_ab.log
