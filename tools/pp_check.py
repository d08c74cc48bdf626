#!/usr/bin/env python3
"""Ping-pong GEMM (ppgemm.hip, opt-in: ETAINV_PP=1; ETAINV_PP_FORCE_ALL=1 also takes the residual / statistics variants, which spill) against the ring kernel on the same inputs: equality and time.  GPU box only.
    python tools/pp_check.py [dualn] [res] [stat]     # dualn: the dual-N kernel (ETAINV_DUALN) instead; which epilogue variants to include"""
import ctypes as C
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
SWITCH = "ETAINV_DUALN" if "dualn" in sys.argv else "ETAINV_PP"
variants = [(False, False)] + ([(True, False)] if "res" in sys.argv else []) + ([(False, True), (True, True)] if "stat" in sys.argv else [])


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


shapes = [(524288, 320, 320), (524288, 320, 1280), (131072, 640, 640), (131072, 640, 2560), (32768, 1280, 1280), (32768, 1280, 5120), (524288, 960, 320),
          (393216, 320, 320), (24576, 1280, 1280)]
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0
for m, n, k in shapes:
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).to(dt)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(dt)
    bias = torch.randn(n, device="cuda", generator=g)
    res_t = (torch.randn(m, n, device="cuda", generator=g) * 0.5).to(dt)
    for with_res, with_stat in variants:
        res = res_t if with_res else None
        outs, stats, ms = {}, {}, {}
        for pp in ("0", "1"):
            os.environ[SWITCH] = pp
            out = torch.full((m, n), float("nan"), dtype=dt, device="cuda")
            part = torch.full((m, 16, 2), float("nan"), device="cuda") if with_stat else None
            P = C.c_int(0)
            if with_stat:
                fn = lambda: _capi.check(lib.etainv_op_gemm_ln(_capi.ptr(x), _capi.ptr(w), _capi.ptr(bias), None, None, _capi.ptr(res), _capi.ptr(out), _capi.ptr(part),
                                                                C.byref(P), m, n, k, 0, code, st))
            else:
                fn = lambda: _capi.check(lib.etainv_op_gemm(_capi.ptr(x), _capi.ptr(w), _capi.ptr(bias), _capi.ptr(res), _capi.ptr(out), m, n, k, 0, code, st))
            fn()
            torch.cuda.synchronize()
            outs[pp] = out
            stats[pp] = part.reshape(-1)[: m * P.value * 2].clone() if with_stat else None
            ms[pp] = min(timeit(fn) for _ in range(3))
        same = torch.equal(outs["0"], outs["1"])
        sdiff = float((stats["0"] - stats["1"]).abs().max()) if with_stat else 0.0
        bad += (not same) or not (sdiff < 1e-5)
        ref = x[:4096].float() @ w.float().t() + bias + (res[:4096].float() if with_res else 0)
        err = float((outs["1"][:4096].float() - ref).norm() / ref.norm())
        fl = 2.0 * m * n * k
        print(f"M={m} N={n} K={k} res={int(with_res)} stat={int(with_stat)}: ring {ms['0']:.3f} ms {fl / ms['0'] / 1e9:7.1f} TF | pp {ms['1']:.3f} ms {fl / ms['1'] / 1e9:7.1f} TF | "
              f"pp/ring {ms['1'] / ms['0']:.3f} | bit-equal {same}" + (f" | stats max diff {sdiff:.2e}" if with_stat else "") + f" | rel err vs fp32 {err:.2e}", flush=True)
print("MISMATCHES:", bad)
