#!/bin/bash
# End-of-round verification on a GPU box (through gpurun): the full -m gpu suite, smoke(), the default bench invocation (with its cpu_baseline leg) and a
# rocprofv3 kernel-statistics pass of one bench step.  Everything lands under gpurun_out/verify/; copy what should be judged into profiles/.
#   gpurun --timeout 3300 -- 'bash tools/verify_gpu.sh'
OUT=gpurun_out/verify
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
(time python -m pytest tests -q -m gpu --durations=25) > $OUT/t_full.log 2>&1; echo "rc $?" >> $OUT/t_full.log   # the driver kills this at 1200 s: keep it under 600
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc $?" >> $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b32 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_b32_under_rocprof.json 2> $OUT/prof_b32.err
find $OUT/prof_b32 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_b32.csv \;
rm -rf $OUT/prof_b32
python tools/unet_call.py --rows 128 --calls 2 --shapes > $OUT/unet_shapes_rows128.log 2>&1
grep -h -E "passed|failed|^rc|real|smoke:|s call|s setup" $OUT/t_full.log $OUT/smoke.log; head -c 400 $OUT/bench_default.json
