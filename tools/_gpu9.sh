mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_e2e_gpu.py -q -m gpu -s > gpurun_out/r03/t_exit.log 2>&1; echo "rc $?" >> gpurun_out/r03/t_exit.log
python tools/parity_s50.py --subjects hip_fp16 hip_bf16 hip_fp32 ref_fp16 ref_bf16 --ref-pairs 1 --out gpurun_out/r03/parity_S50_final2.json > gpurun_out/r03/parity_final2.log 2>&1; echo "rc $?" >> gpurun_out/r03/parity_final2.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_exit.json 2> gpurun_out/r03/bench_b32_exit.err
ETAINV_NO_SRC_EXIT=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_b32_noexit.json 2> gpurun_out/r03/bench_b32_noexit.err
python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/r03/bench_cfg5_b.json 2> gpurun_out/r03/bench_cfg5_b.err
python tools/make_synth_pie.py --out /tmp/pie_synth --n 192 > gpurun_out/r03/eval_synth_b.log 2>&1
python eta-inversion_amd/eval.py --data_path /tmp/pie_synth --output /tmp/pie_out --batch 32 --prec fp16 >> gpurun_out/r03/eval_synth_b.log 2>&1
grep -h -E "passed|failed|^rc|exit after|dead-row|FAILED|edited" gpurun_out/r03/t_exit.log gpurun_out/r03/parity_final2.log gpurun_out/r03/eval_synth_b.log | cut -c1-220
