mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ETAINV_LIB=$PWD/eta-inversion_amd/etainv/lib/libetainv_hip_stamps.so ETAINV_IGEMM_STAMPS=1 python tools/experiments/r04_stamps_conv.py > gpurun_out/r04/s_stamps.log 2>&1
grep -v amdgpu.ids gpurun_out/r04/s_stamps.log
