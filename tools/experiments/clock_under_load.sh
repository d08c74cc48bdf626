#!/bin/bash
# What the shader clock and the socket power are while one kernel family runs back to back (rocm-smi, sampled once a second): is a kernel's time set by cycles or by the power cap?
#   bash tools/experiments/clock_under_load.sh "self-attn N=4096"    (any --only pattern of tools/bench_ops.py)
pat="${1:-self-attn N=4096}"
ETAINV_BENCH_ITERS=4000 python tools/bench_ops.py --rows 128 --only "$pat" > /tmp/clk_op.log 2>&1 &
P=$!
sleep 20
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3 | tr '\n' ' '; echo; sleep 1; done
wait $P
grep -E "ms" /tmp/clk_op.log | tail -2
