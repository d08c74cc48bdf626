mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/n_bench_patch_$i.json 2> gpurun_out/r04/n_bench_patch_$i.err
ETAINV_PATCHCONV=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/n_bench_ring_$i.json 2> gpurun_out/r04/n_bench_ring_$i.err
done
for f in gpurun_out/r04/n_bench_*.json; do python -c "
import json
d=json.load(open('$f')); print('$f', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), round(d['end_to_end_mfma_frac'],4), round(d['config']['tflop_per_image'],2))" || tail -3 ${f%.json}.err; done
