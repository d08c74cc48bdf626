mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for m in 32 16 8 4; do
ETAINV_SPLITK_MAX=$m python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('splitk_max $m', round(d['value'],4), round(d['ms_per_step'],1))"
done
done
