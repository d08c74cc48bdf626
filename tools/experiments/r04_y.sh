cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=eta-inversion_amd/etainv/lib
python tools/ab_ops.py --a $L/libetainv_hip.so --b $L/libetainv_hip_a40contig.so --only self-attn --rounds 3 2>&1 | tail -6
