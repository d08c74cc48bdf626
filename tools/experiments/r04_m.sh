mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$PWD/eta-inversion_amd/etainv/lib
(timeout 600 python tools/ab_ops.py --a $L/libetainv_hip.so --b $L/libetainv_hip_pzero.so --env-a ETAINV_PATCHCONV=1 --env-b ETAINV_PATCHCONV=1 --only "conv3x3" --rows 128 --rounds 2 2>&1 | tail -12) > gpurun_out/r04/m_patch_zero.log 2>&1
cat gpurun_out/r04/m_patch_zero.log
