mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$PWD/eta-inversion_amd/etainv/lib/libetainv_hip.so
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "patch_mode" 2>&1 | tail -25) > gpurun_out/r04/j_patch_test.log 2>&1
(timeout 600 python tools/ab_ops.py --a $L --b $L --env-b ETAINV_PATCHCONV=1 --only conv3x3 --rows 128 --rounds 2 2>&1 | tail -14) > gpurun_out/r04/j_patch_ab.log 2>&1
cat gpurun_out/r04/j_patch_test.log gpurun_out/r04/j_patch_ab.log
