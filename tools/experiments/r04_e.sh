mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIB=eta-inversion_amd/etainv/lib/libetainv_hip.so
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "self_attention" 2>&1 | tail -3) > gpurun_out/r04/e_att_test4.log 2>&1
(ETAINV_A40_NW=8 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "self_attention" 2>&1 | tail -3) > gpurun_out/r04/e_att_test8.log 2>&1
(timeout 600 python tools/ab_ops.py --a $LIB --b $LIB --env-b ETAINV_A40_NW=8 --only attn --rows 128 --rounds 3 2>&1 | tail -12) > gpurun_out/r04/e_att_ab.log 2>&1
cat gpurun_out/r04/e_att_test4.log gpurun_out/r04/e_att_test8.log gpurun_out/r04/e_att_ab.log
