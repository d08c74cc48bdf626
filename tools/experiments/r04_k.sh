mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$PWD/eta-inversion_amd/etainv/lib/libetainv_hip_stamps.so
(echo "--- tap-major ring"; ETAINV_LIB=$L ETAINV_IGEMM_STAMPS=1 python tools/experiments/r04_stamps_conv.py 2>&1 | grep -E "==|stamps"; echo "--- PATCH mode"; ETAINV_PATCHCONV=1 ETAINV_LIB=$L ETAINV_IGEMM_STAMPS=1 python tools/experiments/r04_stamps_conv.py 2>&1 | grep -E "==|stamps") > gpurun_out/r04/k_conv_stamps.log 2>&1
cat gpurun_out/r04/k_conv_stamps.log
