#!/usr/bin/env python3
"""Ping-pong GEMM (ppgemm.hip, ETAINV_PP=1) against the ring kernel on the same inputs: equality and time.  GPU box only."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "eta-inversion_amd"))
import torch  # noqa: E402
from etainv import _capi  # noqa: E402

lib = _capi.load()
st = _capi.stream_ptr()
dt = torch.bfloat16
code = _capi.dtype_code(dt)


def run(x, w, bias, res, out, m, n, k):
    _capi.check(lib.etainv_op_gemm(_capi.ptr(x), _capi.ptr(w), _capi.ptr(bias), _capi.ptr(res), _capi.ptr(out), m, n, k, 0, code, st))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


shapes = [(524288, 320, 320, True), (524288, 320, 1280, True), (131072, 640, 640, True), (131072, 640, 2560, True), (32768, 1280, 1280, True),
          (32768, 1280, 5120, True), (524288, 960, 320, False), (65536, 1280, 5120, False), (49152, 320, 128, True)]
g = torch.Generator(device="cuda").manual_seed(0)
for m, n, k, with_res in shapes:
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(dt)
    bias = torch.randn(n, device="cuda", generator=g)
    res = (torch.randn(m, n, device="cuda", generator=g) * 0.5).to(dt) if with_res else None
    outs, ms = {}, {}
    for pp in ("0", "1"):
        os.environ["ETAINV_PP"] = pp
        out = torch.full((m, n), float("nan"), dtype=dt, device="cuda")
        run(x, w, bias, res, out, m, n, k)
        torch.cuda.synchronize()
        outs[pp] = out
        ms[pp] = min(timeit(lambda: run(x, w, bias, res, out, m, n, k)) for _ in range(3))
    same = torch.equal(outs["0"], outs["1"])
    ref = x[:4096].float() @ w.float().t() + bias + (res[:4096].float() if with_res else 0)
    err = float((outs["1"][:4096].float() - ref).norm() / ref.norm())
    fl = 2.0 * m * n * k
    print(f"M={m} N={n} K={k} res={int(with_res)}: ring {ms['0']:.3f} ms {fl / ms['0'] / 1e9:7.1f} TF | pp {ms['1']:.3f} ms {fl / ms['1'] / 1e9:7.1f} TF | "
          f"pp/ring {ms['1'] / ms['0']:.3f} | bit-equal {same} | rel err vs fp32 {err:.2e}", flush=True)
