"""Run one bench-sized conv repeatedly for >= 2 s (clock settles), then print the stamps build's in-kernel clock."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "eta-inversion_amd"))
from etainv import _capi
lib = _capi.load()
dt = torch.bfloat16; code = _capi.dtype_code(dt); st = _capi.stream_ptr()
R, side, cin, cout = 128, 16, 1280, 1280
x = (torch.randn(R, side, side, cin, device="cuda") * 0.5).to(dt)
w = (torch.randn(cout, 9, cin, device="cuda") * (9 * cin) ** -0.5).to(dt)
bias = torch.randn(cout, device="cuda")
out = torch.empty(R, side, side, cout, dtype=dt, device="cuda")
fn = lambda: _capi.check(lib.etainv_op_conv3x3(_capi.ptr(x), None, cin, 0, _capi.ptr(w), _capi.ptr(bias), None, None, _capi.ptr(out), R, side, side, cout, 1, 0, 9, code, st))
t0 = time.time()
while time.time() - t0 < 3.0:
    fn()
torch.cuda.synchronize()
