mkdir -p gpurun_out/r03
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_ring256.json 2> gpurun_out/r03/bench_ring256.err
ETAINV_RING_MIN_TILES=192 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_ring192.json 2> gpurun_out/r03/bench_ring192.err
ETAINV_RING_MIN_TILES=128 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_ring128.json 2> gpurun_out/r03/bench_ring128.err
python - <<'PY'
import json
for f in ('ring256','ring192','ring128'):
    d=json.load(open(f'gpurun_out/r03/bench_{f}.json')); print(f, d['value'], d['roofline']['achieved'])
PY
