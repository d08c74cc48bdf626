# GPU-side measurement pass; the CPU-only reference-precision floor runs (oracle with 16-bit execution emulated, S = 50, pair 0) share the box
mkdir -p gpurun_out/r03 gpurun_out/parity_cache
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(python tools/parity_s50.py --subjects ref_fp16 ref_bf16 --ref-pairs 1 --max-workers 2 --cache-out gpurun_out/parity_cache --out gpurun_out/r03/parity_S50_ref.json > gpurun_out/r03/parity_ref.log 2>&1; echo "rc $?" >> gpurun_out/r03/parity_ref.log) &
PAR=$!
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --with-nets > gpurun_out/r03/bench_b32.json 2> gpurun_out/r03/bench_b32.err
ETAINV_GN_FOLD=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_gnfold.json 2> gpurun_out/r03/bench_b32_gnfold.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_b32 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03/bench_b32_under_rocprof.json 2> gpurun_out/r03/prof_b32.err
find gpurun_out/r03/prof_b32 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03/kernel_stats_b32.csv \;
find gpurun_out/r03/prof_b32 -type f ! -name "*kernel_stats.csv" -delete
python tools/unet_call.py --rows 128 --calls 2 --shapes --dump gpurun_out/r03/launches_rows128.json > gpurun_out/r03/unet_shapes_rows128.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03/pmc_fetch -- python3 tools/unet_call.py --rows 128 --calls 2 > gpurun_out/r03/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03/pmc_write -- python3 tools/unet_call.py --rows 128 --calls 2 > gpurun_out/r03/pmc_write.log 2>&1
F=$(find gpurun_out/r03/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r03/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W > gpurun_out/r03/pmc_traffic_rows128.json 2> gpurun_out/r03/pmc_traffic.err
python tools/pmc_per_launch.py gpurun_out/r03/launches_rows128.json $F $W > gpurun_out/r03/pmc_per_shape_rows128.json 2> gpurun_out/r03/pmc_per_shape.err
rm -rf gpurun_out/r03/pmc_fetch gpurun_out/r03/pmc_write
python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/r03/bench_cfg5.json 2> gpurun_out/r03/bench_cfg5.err
python tools/make_synth_pie.py --out /tmp/pie_synth --n 96 > gpurun_out/r03/eval_synth.log 2>&1
python eta-inversion_amd/eval.py --data_path /tmp/pie_synth --output /tmp/pie_out --batch 32 --prec fp16 >> gpurun_out/r03/eval_synth.log 2>&1
python eta-inversion_amd/eval.py --data_path /tmp/pie_synth --output /tmp/pie_out2 --batch 32 --prec fp16 --io_threads 0 >> gpurun_out/r03/eval_synth.log 2>&1
wait $PAR
tail -3 gpurun_out/r03/parity_ref.log gpurun_out/r03/eval_synth.log
ls -la gpurun_out/r03
