mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "xsgemm or layernorm_fold" 2>&1 | tail -15) > gpurun_out/r04/a_xs_test.log 2>&1
(timeout 300 python tools/bench_xsgemm.py 2>&1 | tail -12) > gpurun_out/r04/a_xs_bench.log 2>&1
cat gpurun_out/r04/a_xs_test.log gpurun_out/r04/a_xs_bench.log
