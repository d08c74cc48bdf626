import sys, os, torch
sys.path.insert(0, "eta-inversion_amd")
from etainv import _capi
lib = _capi.load()
n_img, heads, res, L = 32, 8, 16, 64
acc = (torch.rand(5, n_img, 2, heads, res * res, 77, device="cuda") ** 4) * 3
x = torch.randn(2 * n_img, 4, L, L, device="cuda")
al = torch.zeros(n_img, 2, 77, device="cuda"); al[:, :, 2] = 1
def t(n=20):
    for _ in range(3): _capi.check(lib.etainv_op_local_blend(_capi.ptr(acc), 5, n_img, heads, res, L, _capi.ptr(x), n_img, _capi.ptr(al), 0.3, _capi.stream_ptr()))
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): _capi.check(lib.etainv_op_local_blend(_capi.ptr(acc), 5, n_img, heads, res, L, _capi.ptr(x), n_img, _capi.ptr(al), 0.3, _capi.stream_ptr()))
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
print("local_blend B=32 split: %.3f ms" % t())
os.environ["ETAINV_BLEND_NOSPLIT"] = "1"
print("local_blend B=32 unsplit (1024 threads): %.3f ms" % t())
