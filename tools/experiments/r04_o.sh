# B-early issue (weight pieces with the activation pieces in window 2): parity of the variant builds, then same-box per-shape A/B
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=eta-inversion_amd/etainv/lib
for v in be1 be2; do
ETAINV_LIB=$PWD/$L/libetainv_hip_$v.so timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/r04/o_kernels_$v.log 2>&1; tail -2 gpurun_out/r04/o_kernels_$v.log
done
python tools/ab_ops.py --a $L/libetainv_hip.so --b $L/libetainv_hip_be1.so --only conv --rounds 3 > gpurun_out/r04/o_ab_be1.log 2>&1
python tools/ab_ops.py --a $L/libetainv_hip.so --b $L/libetainv_hip_be2.so --rounds 3 > gpurun_out/r04/o_ab_be2.log 2>&1
cat gpurun_out/r04/o_ab_be1.log gpurun_out/r04/o_ab_be2.log
