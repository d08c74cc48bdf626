"""How the LayerNorm fold's error grows with the ratio of a row's mean to its deviation (the fold multiplies the RAW rows by the
rounded gamma-scaled weights and subtracts mean * s afterwards; the standalone kernel centres in fp32 first).
    python tools/experiments/ln_fold_mean_sweep.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "eta-inversion_amd"))
import torch
import torch.nn.functional as F
from etainv import _capi
lib = _capi.load()
m, c, n = 4096, 320, 960
g = torch.Generator().manual_seed(0)
w = (torch.randn(n, c, generator=g) * c ** -0.5).cuda()
gamma, beta, bias = (1 + 0.3 * torch.randn(c, generator=g)).cuda(), (0.2 * torch.randn(c, generator=g)).cuda(), torch.randn(n, generator=g).cuda()
for dtype in (torch.float16, torch.bfloat16):
    dt = _capi.dtype_code(dtype)
    wp = torch.empty(n, c, dtype=dtype, device="cuda")
    s_vec, c_vec = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    _capi.check(lib.etainv_op_ln_fold(_capi.ptr(w), _capi.ptr(gamma), _capi.ptr(beta), _capi.ptr(bias), n, c, 0, 1.0, _capi.ptr(wp), _capi.ptr(s_vec), _capi.ptr(c_vec), dt, _capi.stream_ptr()))
    for ratio in (0.0, 0.3, 1.0, 3.0, 10.0, 30.0):
        x = (torch.randn(m, c, generator=g) + ratio).to(dtype).cuda()
        stat = torch.empty(m, 2, device="cuda")
        _capi.check(lib.etainv_op_row_stats(_capi.ptr(x), _capi.ptr(stat), m, c, 1e-5, dt, _capi.stream_ptr()))
        out = torch.empty(m, n, dtype=dtype, device="cuda")
        _capi.check(lib.etainv_op_gemm_ln(_capi.ptr(x), _capi.ptr(wp), _capi.ptr(c_vec), _capi.ptr(s_vec), _capi.ptr(stat), None, _capi.ptr(out), None, None, m, n, c, 0, dt, _capi.stream_ptr()))
        ref = F.layer_norm(x.float(), (c,), gamma, beta, 1e-5) @ w.t() + bias
        # the unfused path for comparison: LayerNorm kernel -> rounded activations -> plain GEMM
        y = torch.empty_like(x)
        _capi.check(lib.etainv_op_layernorm(_capi.ptr(x), _capi.ptr(gamma), _capi.ptr(beta), _capi.ptr(y), m, c, 1e-5, dt, _capi.stream_ptr()))
        wq = w.to(dtype)
        out2 = torch.empty(m, n, dtype=dtype, device="cuda")
        _capi.check(lib.etainv_op_gemm(_capi.ptr(y), _capi.ptr(wq), _capi.ptr(bias), None, _capi.ptr(out2), m, n, c, 0, dt, _capi.stream_ptr()))
        e1 = ((out.float() - ref).norm() / ref.norm()).item()
        e2 = ((out2.float() - ref).norm() / ref.norm()).item()
        print(f"{str(dtype):16s} mean/std {ratio:5.1f}: folded rel L2 {e1:.2e}   standalone LayerNorm + GEMM {e2:.2e}")
