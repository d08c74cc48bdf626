import sys, torch
sys.path.insert(0, "eta-inversion_amd")
from etainv import _capi
lib = _capi.load()
dt = torch.float16; code = _capi.dtype_code(dt); st = _capi.stream_ptr()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for b_, h, cin, cout in ((4, 8, 1280, 1280), (1, 8, 1280, 1280), (4, 8, 2560, 1280), (1, 16, 1280, 1280), (4, 16, 1280, 1280)):
    x = (torch.randn(b_, h, h, cin, device="cuda") * 0.5).to(dt)
    w = (torch.randn(cout, 9, cin, device="cuda") * (9 * cin) ** -0.5).to(dt)
    bias = torch.randn(cout, device="cuda")
    out = torch.empty(b_, h, h, cout, dtype=dt, device="cuda")
    fn = lambda: _capi.check(lib.etainv_op_conv3x3(_capi.ptr(x), None, cin, 0, _capi.ptr(w), _capi.ptr(bias), None, None, _capi.ptr(out), b_, h, h, cout, 1, 0, 9, code, st))
    ms = timeit(fn)
    wbytes = cout * 9 * cin * 2
    print(f"conv {cin}->{cout} @{h} b={b_}: {ms*1e3:8.1f} us  weights {wbytes/1e6:.1f} MB -> {wbytes/ms/1e9:.2f} TB/s weight stream, {2.0*b_*h*h*cout*9*cin/ms/1e9:.0f} TFLOP/s")
