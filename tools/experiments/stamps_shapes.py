"""Diagnostic (stamps build only): per-K-step cycle shares of the ring GEMM on the short-K shapes of the transformer blocks.
    ETAINV_LIB=eta-inversion_amd/etainv/lib/libetainv_hip_stamps.so ETAINV_IGEMM_STAMPS=1 python tools/experiments/stamps_shapes.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "eta-inversion_amd"))
import torch
from etainv import _capi
lib = _capi.load()
dt = torch.bfloat16
code = _capi.dtype_code(dt)
st = _capi.stream_ptr()
R = 128
shapes = [("qkv 320->960 @64", 64, 320, 960, 1, False, 0), ("out 320->320 @64 +res", 64, 320, 320, 1, True, 0), ("ff2 1280->320 @64 +res", 64, 1280, 320, 1, True, 0),
          ("640->640 @32 +res", 32, 640, 640, 1, True, 0), ("1280->1280 @16 +res", 16, 1280, 1280, 1, True, 0), ("conv3x3 320->320 @64", 64, 320, 320, 9, False, 0),
          ("conv3x3 1280->1280 @16", 16, 1280, 1280, 9, False, 0), ("geglu 320->2560 @64", 64, 320, 2560, 1, False, 1)]
for name, side, cin, cout, taps, res, geglu in shapes:
    x = (torch.randn(R, side, side, cin, device="cuda") * 0.5).to(dt)
    w = (torch.randn(cout, taps, cin, device="cuda") * (taps * cin) ** -0.5).to(dt)
    bias = torch.randn(cout, device="cuda")
    resid = (torch.randn(R, side, side, cout, device="cuda") * 0.5).to(dt) if res else None
    out = torch.empty(R, side, side, cout // (2 if geglu else 1), dtype=dt, device="cuda")
    print("==", name, file=sys.stderr, flush=True)
    for _ in range(2):
        if geglu:
            _capi.check(lib.etainv_op_gemm(_capi.ptr(x), _capi.ptr(w), _capi.ptr(bias), None, _capi.ptr(out), R * side * side, cout, cin, 1, code, st))
        else:
            _capi.check(lib.etainv_op_conv3x3(_capi.ptr(x), None, cin, 0, _capi.ptr(w), _capi.ptr(bias), None, _capi.ptr(resid), _capi.ptr(out), R, side, side, cout, 1, 0, taps, code, st))
    torch.cuda.synchronize()
