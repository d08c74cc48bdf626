mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_s50_gpu.py tests/test_realsize_gpu.py -x -q -s > gpurun_out/r04/ae_parity.log 2>&1
grep -v amdgpu gpurun_out/r04/ae_parity.log | tail -40
