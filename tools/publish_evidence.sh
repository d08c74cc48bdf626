#!/bin/bash
# Copy the judged summaries of a closing chain (tools/final_evidence.sh <tag> -> gpurun_out/<tag>/) into profiles/ under the round's prefix.
#   bash tools/publish_evidence.sh r06final9 r06
set -eu
TAG="$1"; PFX="$2"; O=gpurun_out/$TAG
for f in bench_b32_under_rocprof.json bench_cfg2.json bench_cfg5.json bench_default_invocation.json bench_secondary.log bench_torchrun_1rank.json gpu_suite_final.log kernel_stats_b32.csv \
         kernel_stats_rows1.csv kernel_stats_rows4.csv launch_table_rows1.log launch_table_rows128.log launch_table_rows4.log launches_rows128.json ops_rows128_self-attn.log \
         pmc_per_shape_rows128.json pmc_sq_rows128.json pmc_traffic_rows128.json smoke.log unet_shapes_rows128.log unet_shapes_rows32.log unet_shapes_rows32_L96.log; do
  cp "$O/$f" "profiles/${PFX}_$f"
done
echo "published $O -> profiles/${PFX}_*  (stamp $(python tools/srcstamp.py); now take the stamped bench: gpurun -- 'python bench.py > gpurun_out/bench_stamped.json')"
