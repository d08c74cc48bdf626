#!/usr/bin/env python3
"""Time the secondary BASELINE.json configurations on one GPU (synthetic weights / inputs):
   config 2: etainv+simple, 512^2 fp16, 50 steps, batch 1
   config 5: etainv+masactrl, 768^2 fp16, 100 steps, batch 8"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "eta-inversion_amd"))
import torch  # noqa: E402
from etainv.engine import Engine  # noqa: E402
from etainv.pipeline import EtaLoop, noise_table  # noqa: E402


def run(name, L, S, B, editor, dtype, reps=2):
    eng = Engine(dtype=dtype, max_unet_batch=4 * B, latent_size=L, max_img=B)
    eng.load_synthetic(0)
    loop = EtaLoop(eng, S=S, eta=(0.0, 0.4))
    g = torch.Generator().manual_seed(5)
    z0 = (0.9 * torch.randn(B, 4, L, L, generator=g)).cuda()
    cs, ct = torch.randn(B, 2, 77, 768, generator=g).cuda(), torch.randn(B, 2, 77, 768, generator=g).cuda()
    tokens = torch.arange(1, 9, dtype=torch.int32).repeat(B, 1).cuda()
    noise = noise_table(S, 10, L, seed=0)
    masa = (4, 10) if editor == "masactrl" else None

    def step():
        inv = loop.invert(z0, cs, tokens)
        return loop.sample(inv, cs, ct, noise, edit_word=torch.ones(B, dtype=torch.int64), masactrl=masa)
    step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    f_unet = 0.8033 if L == 64 else 2.1481
    tflop = (S * 1 + S * 4) * f_unet
    res = {"config": name, "images_per_s": B / dt, "s_per_batch": dt, "batch": B, "tflop_per_image": tflop,
           "mfma_frac": B / dt * tflop / 2500.0, "finite": bool(torch.isfinite(out).all())}
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    run("2: etainv+simple 512^2 fp16 S=50 B=1", 64, 50, 1, "simple", torch.float16)
    run("5: etainv+masactrl 768^2 fp16 S=100 B=8", 96, 100, 8, "masactrl", torch.float16, reps=1)
