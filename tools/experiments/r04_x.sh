mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r04/x_suite_$i.log 2>&1; tail -3 gpurun_out/r04/x_suite_$i.log | head -1
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
