mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(ETAINV_LIB=$PWD/eta-inversion_amd/etainv/lib/libetainv_hip_a40stamps.so ETAINV_A40_STAMPS=1 timeout 300 python tools/bench_ops.py --only "self-attn N=4096" --rows 128 2>&1 | grep -E "a40 stamps|self-attn" | sort | uniq -c | sort -rn | head -6) > gpurun_out/r04/f_att_stamps.log 2>&1
cat gpurun_out/r04/f_att_stamps.log
