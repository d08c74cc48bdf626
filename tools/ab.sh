#!/bin/bash
# One parametrised runner for the GPU-box measurements (replaces the per-experiment scratch scripts of rounds 3-4).
#   gpurun --timeout T -- 'bash tools/ab.sh <tag> <job> [args] [-- <job> [args] ...]'
# Results go to gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/ and add a row to profiles/README.md.
# Jobs:
#   sq                         SQ counter passes over one 128-row UNet call -> pmc_sq_rows128.json        (tools/pmc_sq.py)
#   traffic                    FETCH_SIZE / WRITE_SIZE passes -> pmc_traffic_rows128.json, pmc_per_shape_rows128.json
#   shapes [rows] [L]          event-timed per-shape breakdown of one UNet call (L = latent side, default 64) -> unet_shapes_rows<rows>[_L<L>].log, launches_*.json
#   routes [rows]              per-(shape, kernel family) table of one UNet call: launch trace joined with event times -> launch_table_rows<rows>.log
#   ops [only]                 per-shape micro-benchmark (tools/bench_ops.py --rows 128)
#   opsenv <only> <envB>       same-box A/B of the default library: plain vs with VAR=value[,VAR=value] (tools/ab_ops.py)
#   pmcop <name> <only> <ctrs..>  one --pmc pass over tools/bench_ops.py --only <only>, per-kernel summary -> pmcop_<name>.json
#   counters                   rocprofv3 -L -> counters.txt
#   opsenv2 <only> <envA> <envB> same-box A/B of the default library under two environments
#   opslib <only> <variant>    same-box A/B: libetainv_hip.so vs libetainv_hip_<variant>.so (built by csrc/build.sh VARIANT=...)
#   bench [args]               python bench.py [args] (default invocation when no args) -> bench<sanitised args>.json
#   benchenv <envB> [n]        bench A/B (--steps 2 --warmup 1 --no-cpu-baseline), n alternations (default 2)
#   stats                      rocprofv3 --kernel-trace --stats of one bench step -> kernel_stats_b32.csv
#   pytest [args]              python -m pytest tests -m gpu -x -q [args] -> gpu_suite.log
#   py <file> [args]           python <file> [args] -> <basename>.log
set -u
TAG="$1"; shift
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIB=eta-inversion_amd/etainv/lib

pyjson() { python -c "
import json,sys
for l in sys.stdin.read().strip().splitlines()[::-1]:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', round(d['value'],4), 'img/s', round(d['ms_per_step'],1), 'ms  igemm', round(d['roofline']['achieved'],1), 'TF  e2e', d.get('end_to_end_mfma_frac'), ' attn', round(d['other_kernels']['self_attention']['tflops'],1)); break
"; }

run_job() {
  local job="$1"; shift
  case "$job" in
    sq)
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_sq1.log 2>&1
      rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_sq2.log 2>&1
      python tools/pmc_sq.py $(find $OUT/pmc_sq1 $OUT/pmc_sq2 -name "*counter_collection.csv") > $OUT/pmc_sq_rows128.json 2> $OUT/pmc_sq.err
      rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2
      tail -2 $OUT/pmc_sq1.log; tail -2 $OUT/pmc_sq2.log; cat $OUT/pmc_sq.err; head -c 1500 $OUT/pmc_sq_rows128.json; echo ;;
    traffic)
      python tools/unet_call.py --rows 128 --calls 2 --shapes --dump $OUT/launches_rows128.json > $OUT/unet_shapes_rows128.log 2>&1
      rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_fetch.log 2>&1
      rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/unet_call.py --rows 128 --calls 2 > $OUT/pmc_write.log 2>&1
      F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
      python tools/pmc_traffic.py $F $W > $OUT/pmc_traffic_rows128.json 2> $OUT/pmc_traffic.err
      python tools/pmc_per_launch.py $OUT/launches_rows128.json $F $W > $OUT/pmc_per_shape_rows128.json 2> $OUT/pmc_per_shape.err
      rm -rf $OUT/pmc_fetch $OUT/pmc_write
      head -c 400 $OUT/pmc_traffic_rows128.json; echo ;;
    shapes)
      local rows="${1:-128}" L="${2:-64}" sfx=""
      [ "$L" != "64" ] && sfx="_L$L"
      python tools/unet_call.py --rows $rows --L $L --calls 2 --shapes --dump $OUT/launches_rows$rows$sfx.json > $OUT/unet_shapes_rows$rows$sfx.log 2>&1
      grep -E "^==|total" $OUT/unet_shapes_rows$rows$sfx.log ;;
    routes)
      local rows="${1:-128}"
      ETAINV_TRACE_IGEMM=1 python tools/unet_call.py --rows $rows --calls 2 --shapes --dump $OUT/launches_routes_rows$rows.json > $OUT/routes_shapes.log 2> $OUT/routes_trace.txt
      python tools/launch_table.py $OUT/launches_routes_rows$rows.json $OUT/routes_trace.txt > $OUT/launch_table_rows$rows.log 2>&1; rm -f $OUT/routes_trace.txt
      cat $OUT/launch_table_rows$rows.log ;;
    ops)
      python tools/bench_ops.py --rows 128 ${1:+--only "$1"} > $OUT/ops_rows128${1:+_$1}.log 2>&1; cat $OUT/ops_rows128${1:+_$1}.log ;;
    opsenv)
      python tools/ab_ops.py --a $LIB/libetainv_hip.so --b $LIB/libetainv_hip.so --only "$1" --env-b "$2" > "$OUT/opsenv_$1_${2//[^A-Za-z0-9_=]/_}.log" 2>&1
      echo "A = default, B = $2"; cat "$OUT/opsenv_$1_${2//[^A-Za-z0-9_=]/_}.log" ;;
    pmcop)
      # pmcop <name> <only> <counter> [counter ...]: one rocprofv3 --pmc pass over the micro-benchmark of one op, aggregated per kernel by tools/pmc_sq.py
      local name="$1" only="$2"; shift 2
      rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmcop_$name -- python3 tools/bench_ops.py --rows 128 --only "$only" > $OUT/pmcop_$name.log 2>&1
      python tools/pmc_sq.py $(find $OUT/pmcop_$name -name "*counter_collection.csv") > $OUT/pmcop_$name.json 2> $OUT/pmcop_$name.err
      rm -rf $OUT/pmcop_$name; tail -3 $OUT/pmcop_$name.log; cat $OUT/pmcop_$name.err
      python -c "
import json,sys
d=json.load(open('$OUT/pmcop_$name.json'))
for k in d['kernels'][:3]:
    print(k['kernel'], k['launches']); print({a:b for a,b in k.items() if a not in ('kernel','raw','launches')}); print(k['raw'])
" ;;
    kstats)
      # kstats <rows> [calls]: rocprofv3 --kernel-trace --stats of tools/unet_call.py at <rows> rows -> kernel_stats_rows<rows>.csv (per-kernel durations without event overhead)
      local rows="${1:-4}" calls="${2:-20}"
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_$rows -- python3 tools/unet_call.py --rows $rows --calls $calls --dtype fp16 > $OUT/kstats_$rows.log 2>&1
      find $OUT/kstats_$rows -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_rows$rows.csv \;
      rm -rf $OUT/kstats_$rows; python - <<PYEOF
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_rows$rows.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernels %d calls: total %.3f ms per call over $calls calls" % (sum(int(r["Calls"]) for r in rows), tot/1e6/$calls))
for r in rows[:28]:
    print("%7.1f us/call-of-unet %6d calls  avg %8.1f us  %5.1f%%  %s" % (float(r["TotalDurationNs"])/1e3/$calls, int(r["Calls"]), float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, r["Name"][:150]))
PYEOF
      ;;
    counters)
      rocprofv3 -L > $OUT/counters.txt 2>&1; grep -c . $OUT/counters.txt ;;
    opsenv2)
      python tools/ab_ops.py --a $LIB/libetainv_hip.so --b $LIB/libetainv_hip.so --only "$1" --env-a "$2" --env-b "$3" > "$OUT/opsenv2_$1_${3//[^A-Za-z0-9_=]/_}.log" 2>&1
      echo "A = $2, B = $3"; cat "$OUT/opsenv2_$1_${3//[^A-Za-z0-9_=]/_}.log" ;;
    opslib)
      python tools/ab_ops.py --a $LIB/libetainv_hip.so --b $LIB/libetainv_hip_$2.so --only "$1" > $OUT/opslib_$1_$2.log 2>&1
      echo "A = default, B = variant $2"; cat $OUT/opslib_$1_$2.log ;;
    bench)
      local name="bench$(echo "$*" | tr -c 'A-Za-z0-9\n' '_')"
      python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; pyjson "$name" < $OUT/$name.json ;;
    benchenv)
      local n="${2:-2}"
      for i in $(seq $n); do
        python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tee $OUT/benchenv_a_$i.json | pyjson "A default"
        env ${1//,/ } python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tee $OUT/benchenv_b_$i.json | pyjson "B $1"
      done ;;
    stats)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b32 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_b32_under_rocprof.json 2> $OUT/prof_b32.err
      find $OUT/prof_b32 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_b32.csv \;
      rm -rf $OUT/prof_b32; head -8 $OUT/kernel_stats_b32.csv ;;
    pytest)
      timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $OUT/gpu_suite.log 2>&1; tail -5 $OUT/gpu_suite.log ;;
    py)
      local f="$1"; shift
      python "$f" "$@" > $OUT/$(basename "$f" .py).log 2>&1; tail -40 $OUT/$(basename "$f" .py).log ;;
    *) echo "unknown job $job"; return 1 ;;
  esac
}

args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then
    [ ${#args[@]} -gt 0 ] && { echo "=== ${args[*]}"; run_job "${args[@]}"; }
    args=()
  else
    args+=("$a")
  fi
done
[ ${#args[@]} -gt 0 ] && { echo "=== ${args[*]}"; run_job "${args[@]}"; }
exit 0
