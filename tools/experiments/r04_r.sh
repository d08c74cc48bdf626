mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_configs_gpu.py -x -q -s --durations=5 > gpurun_out/r04/r_cfg4.log 2>&1
tail -15 gpurun_out/r04/r_cfg4.log
