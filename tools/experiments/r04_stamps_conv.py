"""Diagnostic (stamps build only): per-K-step cycle shares of the conv3x3 ring kernel, tap-major tiles vs PATCH mode (ETAINV_PATCHCONV).
    ETAINV_LIB=.../libetainv_hip_stamps.so ETAINV_IGEMM_STAMPS=1 [ETAINV_PATCHCONV=1] python tools/experiments/r04_stamps_conv.py"""
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
for name, side, cin, cout in (("conv3x3 640->320 @64", 64, 640, 320), ("conv3x3 640->640 @32", 32, 640, 640), ("conv3x3 1280->1280 @16", 16, 1280, 1280)):
    x = (torch.randn(R, side, side, cin, device="cuda") * 0.5).to(dt)
    w = (torch.randn(cout, 9, cin, device="cuda") * (9 * cin) ** -0.5).to(dt)
    bias = torch.randn(cout, device="cuda")
    out = torch.empty(R, side, side, cout, dtype=dt, device="cuda")
    print("==", name, file=sys.stderr, flush=True)
    for _ in range(2):
        _capi.check(lib.etainv_op_conv3x3(_capi.ptr(x), None, cin, 0, _capi.ptr(w), _capi.ptr(bias), None, None, _capi.ptr(out), R, side, side, cout, 1, 0, 9, code, st))
    torch.cuda.synchronize()
