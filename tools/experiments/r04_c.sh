mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(time timeout 1500 python -m pytest tests -q -m gpu --durations=40) > gpurun_out/r04/c_full.log 2>&1; echo "rc $?" >> gpurun_out/r04/c_full.log
(time python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r04/c_smoke.log 2>&1; echo "rc $?" >> gpurun_out/r04/c_smoke.log
tail -60 gpurun_out/r04/c_full.log; tail -5 gpurun_out/r04/c_smoke.log
