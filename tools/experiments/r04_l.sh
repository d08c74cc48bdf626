mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$PWD/eta-inversion_amd/etainv/lib
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "patch_mode or conv3x3" 2>&1 | tail -3) > gpurun_out/r04/l_patch_test.log 2>&1
(timeout 600 python tools/ab_ops.py --a $L/libetainv_hip.so --b $L/libetainv_hip.so --env-a ETAINV_PATCHCONV=0 --only conv3x3 --rows 128 --rounds 2 2>&1 | tail -12) > gpurun_out/r04/l_patch_ab.log 2>&1
(ETAINV_LIB=$L/libetainv_hip_stamps.so ETAINV_IGEMM_STAMPS=1 python tools/experiments/r04_stamps_conv.py 2>&1 | grep -E "==|stamps 256") > gpurun_out/r04/l_patch_stamps.log 2>&1
cat gpurun_out/r04/l_patch_test.log gpurun_out/r04/l_patch_ab.log gpurun_out/r04/l_patch_stamps.log
