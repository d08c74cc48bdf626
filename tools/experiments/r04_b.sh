mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "xsgemm" 2>&1 | tail -5) > gpurun_out/r04/b_xs_test.log 2>&1
(ETAINV_LIB=$PWD/eta-inversion_amd/etainv/lib/libetainv_hip_xsstamps.so ETAINV_XS_STAMPS=1 timeout 300 python tools/bench_xsgemm.py --rows 128 2>&1 | grep "xs stamps" | sort | uniq -c | sort -rn | head -8) > gpurun_out/r04/b_xs_stamps.log 2>&1
(timeout 300 python tools/bench_xsgemm.py --rows 128 2>&1 | tail -3) > gpurun_out/r04/b_xs_bench.log 2>&1
cat gpurun_out/r04/b_xs_test.log gpurun_out/r04/b_xs_stamps.log gpurun_out/r04/b_xs_bench.log
