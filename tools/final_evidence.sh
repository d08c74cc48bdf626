#!/bin/bash
# The round's closing measurement chain on one GPU box (about 10 minutes): everything profiles/README.md's "final kernels" rows cite.
#   gpurun --timeout 3000 -- 'bash tools/final_evidence.sh <tag>'     -> gpurun_out/<tag>/ ; copy into profiles/ as rNN_*
set -u
TAG="${1:-final}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_default_invocation.json 2> $O/bench_default.err
bash tools/ab.sh $TAG stats -- shapes 128 -- routes 128 -- traffic -- sq
# the single-image path (BASELINE config 2: UNet calls of 1 .. 4 rows) and config 5's shapes (32 rows at L = 96): per-launch tables and kernel-level durations
bash tools/ab.sh $TAG routes 4 -- routes 1 -- kstats 4 20 -- kstats 1 20 -- shapes 32 96 -- shapes 32 -- ops self-attn
python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null
python bench.py --config 2 --no-cpu-baseline > $O/bench_cfg2.json 2>/dev/null
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_again.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun.err
python - <<PY > $O/bench_secondary.log
import json
for n in ("bench_default_invocation", "bench_cfg5", "bench_cfg2", "bench_cfg3_again", "bench_torchrun_1rank"):
    for l in open("$O/%s.json" % n).read().strip().splitlines()[::-1]:
        try: d = json.loads(l)
        except Exception: continue
        print(n, d["metric"], round(d["value"], 4), d["unit"], "ms_per_step", round(d["ms_per_step"], 1), "igemm", round(d["roofline"]["achieved"], 1), "TF e2e", d.get("end_to_end_mfma_frac")); break
PY
cat $O/bench_secondary.log
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite_final.log 2>&1; tail -3 $O/gpu_suite_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
