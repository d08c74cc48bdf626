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
a = ap.parse_args()
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
B = a.rows // 4
eng = Engine(dtype=dt, max_unet_batch=a.rows, latent_size=64, max_img=B)
eng.load_synthetic(0)
g = torch.Generator().manual_seed(0)
x = torch.randn(2 * B, 4, 64, 64, generator=g).cuda()
ctx = torch.randn(a.rows, 77, 768, generator=g).cuda()
out = torch.empty(a.rows, 4, 64, 64, device="cuda")
for i in range(a.calls):
    eng.unet(x, 500, ctx, None, out=out)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
