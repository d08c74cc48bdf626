#!/usr/bin/env python3
"""Run a few UNet calls of the bench workload shape (no event profiler) -- target for rocprofv3 --pmc passes."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "eta-inversion_amd"))
import torch  # noqa: E402
from etainv.engine import Engine, AttnControl  # noqa: E402
from etainv import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=128)
ap.add_argument("--calls", type=int, default=2)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--L", type=int, default=64, help="latent side (64 = 512^2 images, 96 = 768^2: BASELINE config 5)")
ap.add_argument("--shapes", action="store_true", help="event-time every launch of the last call and print time per distinct work size")
ap.add_argument("--dump", default=None, help="with --shapes: write the igemm launches of the last call in launch order (ms, FLOPs, algorithmic bytes) as JSON")
a = ap.parse_args()
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
B = a.rows // 4
eng = Engine(dtype=dt, max_unet_batch=a.rows, latent_size=a.L, max_img=max(B, 1))
eng.load_synthetic(0)
g = torch.Generator().manual_seed(0)
x = torch.randn(max(2 * B, 1) if a.rows >= 4 else a.rows, 4, a.L, a.L, generator=g).cuda()
ctx = torch.randn(a.rows, 77, 768, generator=g).cuda()
out = torch.empty(a.rows, 4, a.L, a.L, device="cuda")
lib = _capi.load()
for i in range(a.calls):
    if a.shapes and i == a.calls - 1:
        lib.etainv_prof_enable(1)
        lib.etainv_prof_reset()
    eng.unet(x, 500, ctx, None, out=out)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
if a.shapes:
    import ctypes as C
    from collections import defaultdict
    names = ["igemm (FLOP)", "self-attn (FLOP)", "cross-attn (FLOP)", "groupnorm (B)", "layernorm (B)"]
    total = 0.0
    for cls, nm in enumerate(names):
        cap = 4096
        ms, work, n = (C.c_double * cap)(), (C.c_double * cap)(), C.c_int64(0)
        _capi.check(lib.etainv_prof_records(cls, ms, work, cap, C.byref(n)))
        agg = defaultdict(lambda: [0, 0.0])
        for i in range(min(n.value, cap)):
            agg[work[i]][0] += 1
            agg[work[i]][1] += ms[i]
        t = sum(v[1] for v in agg.values())
        total += t
        print(f"== {nm}: {n.value} launches, {t:.2f} ms")
        for w, (cnt, tm) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"   work {w:14.4g}  x{cnt:3d}  {tm:8.3f} ms  ({tm / cnt:7.3f} each)  {w * cnt / tm / 1e9:9.1f} G/s")
    print(f"total event-timed: {total:.2f} ms")
    if a.dump:
        import json
        cap = 4096
        ms, work, nbytes, n = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_double * cap)(), C.c_int64(0)
        _capi.check(lib.etainv_prof_records_ex(0, ms, work, nbytes, cap, C.byref(n)))
        json.dump({"rows": a.rows, "dtype": a.dtype, "igemm": [{"ms": ms[i], "flops": work[i], "bytes": nbytes[i]} for i in range(min(n.value, cap))]},
                  open(a.dump, "w"))
