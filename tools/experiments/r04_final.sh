# Round-4 evidence on one box: default bench (with cpu_baseline), rocprofv3 kernel statistics, per-shape times + launch list, PMC fabric traffic
# (FETCH_SIZE / WRITE_SIZE in SEPARATE passes, kernel-trace only), engine build time, secondary configs.  Copy what should be judged into profiles/.
OUT=gpurun_out/r04f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b32 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_b32_under_rocprof.json 2> $OUT/prof_b32.err
find $OUT/prof_b32 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_b32.csv \;
rm -rf $OUT/prof_b32
python tools/unet_call.py --rows 128 --calls 2 --shapes --dump $OUT/launches_rows128.json > $OUT/unet_shapes_rows128.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_write.log 2>&1
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W > $OUT/pmc_traffic_rows128.json 2> $OUT/pmc_traffic.err
python tools/pmc_per_launch.py $OUT/launches_rows128.json $F $W > $OUT/pmc_per_shape_rows128.json 2> $OUT/pmc_per_shape.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
OMP_NUM_THREADS=8 python - > $OUT/engine_build_time.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "eta-inversion_amd")
import torch
from etainv.engine import Engine
t0 = time.time(); e = Engine(dtype=torch.bfloat16, max_unet_batch=128, latent_size=64, max_img=32); t1 = time.time(); e.load_default(0); torch.cuda.synchronize(); t2 = time.time()
print(f"OMP_NUM_THREADS=8: Engine() {t1 - t0:.1f} s, load_default(0) (860 M synthetic parameters drawn on the host, uploaded, packed) {t2 - t1:.1f} s")
PY
python bench.py --config 2 --steps 3 --warmup 1 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --config 5 --steps 2 --warmup 1 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python bench.py --all-rows --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_allrows.json 2> $OUT/bench_allrows.err
head -c 600 $OUT/bench_default.json; echo; cat $OUT/engine_build_time.log; tail -3 $OUT/unet_shapes_rows128.log; head -c 300 $OUT/pmc_traffic_rows128.json
