mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_modules_api_gpu.py tests/test_graph_gpu.py tests/test_batch_gpu.py -x -q -s > gpurun_out/r04/q_attnres.log 2>&1
grep -E "half:|up:|mid:|down_bwd:|passed|failed|Error" gpurun_out/r04/q_attnres.log | tail -20
