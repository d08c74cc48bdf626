cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('phase', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), round(d['config']['tflop_per_image'],2))"
ETAINV_UPS4=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('9tap-2slot', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['achieved'],1), round(d['config']['tflop_per_image'],2))"
done
python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 phase', round(d['value'],4), round(d['ms_per_step'],1))"
python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['value'],4), round(d['ms_per_step'],1))"
