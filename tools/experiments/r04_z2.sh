mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -k head_major > gpurun_out/r04/z_hm_test.log 2>&1; tail -6 gpurun_out/r04/z_hm_test.log
for i in 1 2; do
ETAINV_QKV_HM=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hm', round(d['value'],4), round(d['ms_per_step'],1), round(d['other_kernels']['self_attention']['tflops'],1))"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rm', round(d['value'],4), round(d['ms_per_step'],1), round(d['other_kernels']['self_attention']['tflops'],1))"
done
ETAINV_QKV_HM=1 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 hm', round(d['value'],4), round(d['ms_per_step'],1))"
python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 rm', round(d['value'],4), round(d['ms_per_step'],1))"
