mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -k head_major > gpurun_out/r04/z_hm_test.log 2>&1; tail -12 gpurun_out/r04/z_hm_test.log
for i in 1 2; do
ETAINV_QKV_HM=1 python tools/unet_call.py --rows 128 --calls 3 --shapes > gpurun_out/r04/z_shapes_hm_$i.log 2>&1
python tools/unet_call.py --rows 128 --calls 3 --shapes > gpurun_out/r04/z_shapes_rm_$i.log 2>&1
done
for f in gpurun_out/r04/z_shapes_*.log; do echo $f; grep -E "== igemm|== self-attn|total event|2.749e\+12|3.436e\+11|3.221e\+11  x" $f; done
