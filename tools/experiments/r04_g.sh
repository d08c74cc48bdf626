mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=$PWD/eta-inversion_amd/etainv/lib
for v in "" _a40noload _a40nols; do
  echo "variant '$v'"; ETAINV_LIB=$L/libetainv_hip$v.so timeout 300 python tools/bench_ops.py --only "self-attn N=4096" --rows 128 2>&1 | grep "self-attn"
done > gpurun_out/r04/g_att_noload.log 2>&1
cat gpurun_out/r04/g_att_noload.log
