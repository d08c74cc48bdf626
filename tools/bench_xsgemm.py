#!/usr/bin/env python3
"""A/B of the K = 320 LayerNorm-consumer projections (GEGLU ff.net.0, fused QKV) on csrc/xsgemm.hip vs the ring kernel of csrc/igemm.hip, same box,
same process (ETAINV_XSGEMM is read per launch).  GPU box only.
    python tools/bench_xsgemm.py [--rows 128] [--dtype bf16]
(round 6: xsgemm.hip is in the EXPERIMENTS=1 library only: ETAINV_LIB=eta-inversion_amd/etainv/lib/libetainv_hip_experiments.so python tools/bench_xsgemm.py)"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "eta-inversion_amd"))
import torch  # noqa: E402
from etainv import _capi  # noqa: E402


def timeit(fn, iters=20, warm=5):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="*", default=[128, 96, 32])
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    code = _capi.dtype_code(dt)
    lib = _capi.load()
    st = _capi.stream_ptr()
    c = 320
    for rows in a.rows:
        m = rows * 4096
        x = (torch.randn(m, c, device="cuda") * 0.8 + 0.2).to(dt)
        xf = x.float()
        stat = torch.stack([xf.mean(-1), (xf.var(-1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
        del xf
        for name, n_out, geglu in (("geglu ff1 320->2560", 2560, 1), ("qkv 320->960", 960, 0)):
            w = torch.randn(n_out, c, device="cuda") * c ** -0.5
            gamma, beta, bias = 1.0 + 0.3 * torch.randn(c, device="cuda"), 0.2 * torch.randn(c, device="cuda"), torch.randn(n_out, device="cuda")
            wp = torch.empty(n_out, c, dtype=dt, device="cuda")
            s_vec, c_vec = torch.empty(n_out, device="cuda"), torch.empty(n_out, device="cuda")
            _capi.check(lib.etainv_op_ln_fold(_capi.ptr(w), _capi.ptr(gamma), _capi.ptr(beta), _capi.ptr(bias), n_out, c, geglu, 1.0, _capi.ptr(wp),
                                              _capi.ptr(s_vec), _capi.ptr(c_vec), code, st))
            out = torch.empty(m, n_out // 2 if geglu else n_out, dtype=dt, device="cuda")
            fn = lambda: _capi.check(lib.etainv_op_gemm_ln(_capi.ptr(x), _capi.ptr(wp), _capi.ptr(c_vec), _capi.ptr(s_vec), _capi.ptr(stat), None,
                                                           _capi.ptr(out), None, None, m, n_out, c, geglu, code, st))
            fl = 2.0 * m * n_out * c
            res = {}
            for tag, on in (("xsgemm", "1"), ("ring", "0"), ("xsgemm", "1"), ("ring", "0")):
                os.environ["ETAINV_XSGEMM"] = on
                res.setdefault(tag, []).append(timeit(fn))
            xs, rg = min(res["xsgemm"]), min(res["ring"])
            print(f"rows={rows:4d} {name:22s} xsgemm {xs:7.3f} ms {fl / xs / 1e9:7.1f} TFLOP/s | ring {rg:7.3f} ms {fl / rg / 1e9:7.1f} TFLOP/s | x{rg / xs:5.2f}", flush=True)
    os.environ.pop("ETAINV_XSGEMM", None)


if __name__ == "__main__":
    main()
