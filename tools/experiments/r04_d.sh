mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_kernels_gpu.py -q -x -k "graph or xsgemm or time" 2>&1 | tail -15) > gpurun_out/r04/d_graph_test.log 2>&1
python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04/d_cfg2_graph.json 2> gpurun_out/r04/d_cfg2_graph.err
ETAINV_NO_GRAPH=1 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04/d_cfg2_nograph.json 2> gpurun_out/r04/d_cfg2_nograph.err
cat gpurun_out/r04/d_graph_test.log; for f in gpurun_out/r04/d_cfg2_graph gpurun_out/r04/d_cfg2_nograph; do python -c "
import json,sys
d=json.load(open('$f.json')); print('$f', d['value'], d['ms_per_step'])" || tail -5 $f.err; done
