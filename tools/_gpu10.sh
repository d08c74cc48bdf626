mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(time python -m pytest tests -q -m gpu) > gpurun_out/r03/t_full.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.log 2>&1; echo "rc $?" >> gpurun_out/r03/smoke.log
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_b32 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_b32_under_rocprof.json 2> gpurun_out/r03/prof_b32.err
find gpurun_out/r03/prof_b32 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03/kernel_stats_b32.csv \;
rm -rf gpurun_out/r03/prof_b32
grep -h -E "passed|failed|^rc|real|smoke:" gpurun_out/r03/t_full.log gpurun_out/r03/smoke.log; head -c 600 gpurun_out/r03/bench_default.json
