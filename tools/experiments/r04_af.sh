# PMC fabric traffic of one 128-row UNet call on the round's last kernels (FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only)
OUT=gpurun_out/r04f2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/unet_call.py --rows 128 --calls 2 --shapes --dump $OUT/launches_rows128.json > $OUT/unet_shapes_rows128.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_write.log 2>&1
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W > $OUT/pmc_traffic_rows128.json 2> $OUT/pmc_traffic.err
python tools/pmc_per_launch.py $OUT/launches_rows128.json $F $W > $OUT/pmc_per_shape_rows128.json 2> $OUT/pmc_per_shape.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
python -c "
import json
d=json.load(open('$OUT/pmc_traffic_rows128.json')); print({k:(v['launches'], round(v['hbm_bytes_per_launch']/1e6,1)) for k,v in d.items()})
p=json.load(open('$OUT/pmc_per_shape_rows128.json')); print(p['igemm_launches'], p['measured_bytes']/1e9, p['algorithmic_bytes']/1e9, p['measured_over_algorithmic'])"
