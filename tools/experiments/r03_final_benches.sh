mkdir -p gpurun_out/r03
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --with-nets > gpurun_out/r03/bench_b32_final_nets.json 2> gpurun_out/r03/bench_b32_final_nets.err
python bench.py --all-rows --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_final_allrows.json 2> gpurun_out/r03/bench_b32_final_allrows.err
python bench.py --config 2 --steps 3 --warmup 1 > gpurun_out/r03/bench_cfg2_final.json 2> gpurun_out/r03/bench_cfg2_final.err
python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/r03/bench_cfg5_final.json 2> gpurun_out/r03/bench_cfg5_final.err
python bench.py --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_final_fp16.json 2> gpurun_out/r03/bench_b32_final_fp16.err
ls -la gpurun_out/r03/*final*
