#!/usr/bin/env python3
"""Timing-only ablations of the ping-pong GEMM vs the ring (ETAINV_IGEMM_DEBUG bits: 1 no DMA in the loop, 2 no epilogue, 4 no MFMA; 256 no fragment reads existed in the round-5 prototype that produced profiles/r05_pp_ablation.log).
(round 6: pp_gemm_kernel is in the EXPERIMENTS=1 library only: run with ETAINV_LIB=eta-inversion_amd/etainv/lib/libetainv_hip_experiments.so)"""
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "eta-inversion_amd"))
import torch  # noqa: E402
from etainv import _capi  # noqa: E402
lib = _capi.load(); st = _capi.stream_ptr(); dt = torch.bfloat16; code = _capi.dtype_code(dt)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
g = torch.Generator(device="cuda").manual_seed(0)
for m, n, k in ((65536, 1280, 5120), (131072, 640, 640), (524288, 320, 320)):
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(dt)
    bias = torch.randn(n, device="cuda", generator=g)
    out = torch.empty(m, n, dtype=dt, device="cuda")
    fn = lambda: _capi.check(lib.etainv_op_gemm(_capi.ptr(x), _capi.ptr(w), _capi.ptr(bias), None, _capi.ptr(out), m, n, k, 0, code, st))
    nk = k // 64
    steps = (m // 256) * (n // 160) * nk / 256.0      # K steps per CU
    for pp in ("0", "1"):
        for dbg, name in ((0, "full"), (2, "no epilogue"), (1 | 2, "no DMA, no epilogue"), (4 | 2, "no MFMA, no epilogue"), (1 | 2 | 4, "reads + barriers only"),
                          (256 | 2, "no reads, no epilogue"), (256 | 1 | 2, "MFMA + barriers only")):
            if pp == "0" and dbg & 256:
                continue
            os.environ["ETAINV_PP"] = pp
            os.environ["ETAINV_IGEMM_DEBUG"] = str(dbg)
            ms = min(timeit(fn) for _ in range(3))
            print(f"M={m} N={n} K={k} {'pp  ' if pp == '1' else 'ring'} {name:26s} {ms:7.3f} ms  = {ms * 1e-3 / steps * 2.1e9:7.0f} cycles per K step at 2.1 GHz", flush=True)
