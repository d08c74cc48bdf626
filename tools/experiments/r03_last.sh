OUT=gpurun_out/verify2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc $?" >> $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b32 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_b32_under_rocprof.json 2> $OUT/prof_b32.err
find $OUT/prof_b32 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_b32.csv \;
rm -rf $OUT/prof_b32
tail -2 $OUT/smoke.log; head -c 300 $OUT/bench_default.json
