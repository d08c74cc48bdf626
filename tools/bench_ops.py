import os
#!/usr/bin/env python3
"""Micro-benchmark of the hot kernels at the shapes of one UNet call (rows = UNet batch rows).  GPU box only.
    python tools/bench_ops.py [--rows 64] [--dtype bf16]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "eta-inversion_amd"))
import torch  # noqa: E402
from etainv import _capi  # noqa: E402


def timeit(fn, iters=int(os.environ.get("ETAINV_BENCH_ITERS", "10")), warm=3):
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
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    code = _capi.dtype_code(dt)
    lib = _capi.load()
    R = a.rows
    st = _capi.stream_ptr()
    rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
    total_ms = total_fl = 0.0
    # (name, side, cin, cout, taps, count per forward)
    convs = [("conv3x3 320->320 @64", 64, 320, 320, 9, 5), ("conv3x3 640->320 @64", 64, 640, 320, 9, 2), ("conv3x3 960->320 @64", 64, 960, 320, 9, 1),
             ("conv3x3 640->640 @32", 32, 640, 640, 9, 7), ("conv3x3 1280->640 @32", 32, 1280, 640, 9, 1), ("conv3x3 1920->640 @32", 32, 1920, 640, 9, 1),
             ("conv3x3 1280->1280 @16", 16, 1280, 1280, 9, 8), ("conv3x3 2560->1280 @16", 16, 2560, 1280, 9, 2),
             ("conv3x3 1280->1280 @8", 8, 1280, 1280, 9, 9), ("conv3x3 2560->1280 @8", 8, 2560, 1280, 9, 3),
             ("lin qkv 320->960 @64", 64, 320, 960, 1, 5), ("lin out 320->320 @64", 64, 320, 320, 1, 25), ("lin ff2 1280->320 @64", 64, 1280, 320, 1, 5),
             ("lin qkv 640->1920 @32", 32, 640, 1920, 1, 5), ("lin 640->640 @32", 32, 640, 640, 1, 25), ("lin ff2 2560->640 @32", 32, 2560, 640, 1, 5),
             ("lin qkv 1280->3840 @16", 16, 1280, 3840, 1, 5), ("lin 1280->1280 @16", 16, 1280, 1280, 1, 25), ("lin ff2 5120->1280 @16", 16, 5120, 1280, 1, 5)]
    print(f"rows={R} dtype={a.dtype}")
    for name, side, cin, cout, taps, cnt in convs:
        if a.only and a.only not in name:
            continue
        x = rnd(R, side, side, cin)
        w = rnd(cout, taps, cin) * (taps * cin) ** -0.5
        # as in the UNet: qkv projections have no bias; out-proj / ff2 / proj_out add bias and the residual stream
        bias = None if "qkv" in name else torch.randn(cout, device="cuda")
        resid = rnd(R, side, side, cout) if (taps == 1 and "qkv" not in name) else None
        out = torch.empty(R, side, side, cout, dtype=dt, device="cuda")
        fn = lambda: _capi.check(lib.etainv_op_conv3x3(_capi.ptr(x), None, cin, 0, _capi.ptr(w), _capi.ptr(bias), None, _capi.ptr(resid), _capi.ptr(out),
                                                        R, side, side, cout, 1, 0, taps, code, st))
        ms = timeit(fn)
        fl = 2.0 * R * side * side * cout * taps * cin
        total_ms += ms * cnt
        total_fl += fl * cnt
        print(f"{name:28s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s   x{cnt}")
    for c, side, cnt in ((320, 64, 5), (640, 32, 5), (1280, 16, 5)):
        name = f"geglu ff1 {c}->{8 * c} @{side}"
        if a.only and a.only not in name:
            continue
        m = R * side * side
        x, w = rnd(m, c), rnd(8 * c, c) * c ** -0.5
        bias = torch.randn(8 * c, device="cuda")
        out = torch.empty(m, 4 * c, dtype=dt, device="cuda")
        fn = lambda: _capi.check(lib.etainv_op_gemm(_capi.ptr(x), _capi.ptr(w), _capi.ptr(bias), None, _capi.ptr(out), m, 8 * c, c, 1, code, st))
        ms = timeit(fn)
        fl = 2.0 * m * 8 * c * c
        total_ms += ms * cnt
        total_fl += fl * cnt
        print(f"{name:28s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s   x{cnt}")
    if total_ms:
        print(f"igemm weighted: {total_ms:.2f} ms per UNet call, {total_fl / total_ms / 1e9:.1f} TFLOP/s")
    for n, d, cnt in ((4096, 40, 5), (9216, 40, 0), (1024, 80, 5), (256, 160, 5)):
        name = f"self-attn N={n} d={d}"
        if a.only and a.only not in name:
            continue
        Rn = R if n <= 4096 else max(1, R // 4)
        qkv = rnd(Rn, n, 3 * 8 * d)
        out = torch.empty(Rn, n, 8 * d, dtype=dt, device="cuda")
        fn = lambda: _capi.check(lib.etainv_op_self_attention(_capi.ptr(qkv), _capi.ptr(out), Rn, n, 8, d, 0, 1, code, st))
        ms = timeit(fn)
        fl = 4.0 * Rn * 8 * n * n * d
        print(f"{name:28s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s   x{cnt}")
    import ctypes as C
    for n, d in ((4096, 40), (1024, 80), (256, 160)):
        for tag in ("plain", "ptp-edit"):
            name = f"cross-attn N={n} d={d} {tag}"
            if a.only and a.only not in name:
                continue
            n_img = R // 4
            q = rnd(R, n, 8 * d)
            kv = rnd(R, 77, 16 * d)
            out = torch.empty(R, n, 8 * d, dtype=dt, device="cuda")
            ctrl = None
            if tag == "ptp-edit":
                mapper = torch.arange(77, dtype=torch.int32, device="cuda").repeat(n_img, 1).contiguous()
                ones = torch.ones(n_img, 77, device="cuda")
                ctrl = _capi.AttnCtrl(mode=_capi.ATTN_PTP, n_img=n_img, store_maps=0, mapper=_capi.ptr(mapper), alphas=_capi.ptr(ones),
                                      equalizer=_capi.ptr(ones), cross_alpha=_capi.ptr(ones))
            fn = lambda: _capi.check(lib.etainv_op_cross_attention(_capi.ptr(q), _capi.ptr(kv), _capi.ptr(out), R, n, 8, d, 77,
                                                                   C.byref(ctrl) if ctrl is not None else None, -1, n_img, None, code, st))
            ms = timeit(fn)
            print(f"{name:34s} {ms:8.3f} ms  {4.0 * R * 8 * n * 77 * d / ms / 1e9:8.1f} TFLOP/s")
    for c, hw in ((320, 4096), (640, 1024), (640, 4096), (960, 4096), (1280, 256), (1280, 64), (2560, 256)):
        name = f"groupnorm C={c} hw={hw}"
        if a.only and a.only not in name:
            continue
        x = rnd(R, hw, c)
        g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        out = torch.empty_like(x)
        scratch = torch.zeros(R * 65 * 64, device="cuda")
        fn = lambda: _capi.check(lib.etainv_op_groupnorm(_capi.ptr(x), None, c, 0, _capi.ptr(g), _capi.ptr(b), _capi.ptr(out), R, hw, 32, 1e-5, 1,
                                                         _capi.ptr(scratch), code, st))
        ms = timeit(fn)
        print(f"{name:28s} {ms:8.3f} ms  {4.0 * R * hw * c / ms / 1e6:8.1f} GB/s (algorithmic r+w)")


if __name__ == "__main__":
    main()
