mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=eta-inversion_amd/etainv/lib
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > gpurun_out/r04/t_kernels.log 2>&1; tail -2 gpurun_out/r04/t_kernels.log
ETAINV_LIB=$PWD/$L/libetainv_hip_stamps.so ETAINV_IGEMM_STAMPS=1 python tools/experiments/r04_stamps_conv.py > gpurun_out/r04/t_stamps.log 2>&1
grep -v amdgpu.ids gpurun_out/r04/t_stamps.log | grep -v "clock"
python tools/ab_ops.py --a $L/libetainv_hip_prev.so --b $L/libetainv_hip.so --only conv --rounds 3 > gpurun_out/r04/t_ab.log 2>&1
cat gpurun_out/r04/t_ab.log
