mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$PWD/eta-inversion_amd/etainv/lib
(timeout 600 python tools/ab_ops.py --a $L/libetainv_hip_nob.so --b $L/libetainv_hip_noab.so --only conv3x3 --rows 128 --rounds 2 2>&1 | tail -14) > gpurun_out/r04/i_conv_nob.log 2>&1
cat gpurun_out/r04/i_conv_nob.log
