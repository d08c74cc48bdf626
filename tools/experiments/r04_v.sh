mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_cfg2 -- python3 bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/v_cfg2.json 2> gpurun_out/r04/v_cfg2.err
find gpurun_out/r04/prof_cfg2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04/v_kernel_stats_cfg2.csv \;
rm -rf gpurun_out/r04/prof_cfg2
head -c 300 gpurun_out/r04/v_cfg2.json; echo
python tools/unet_call.py --rows 2 --calls 3 --shapes --dtype fp16 > gpurun_out/r04/v_shapes_rows2.log 2>&1
python tools/unet_call.py --rows 4 --calls 3 --shapes --dtype fp16 > gpurun_out/r04/v_shapes_rows4.log 2>&1
tail -5 gpurun_out/r04/v_shapes_rows4.log
