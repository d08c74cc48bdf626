#!/usr/bin/env python3
"""Ping-pong GEMM (ppgemm.hip, opt-in: ETAINV_PP=1; ETAINV_PP_FORCE_ALL=1 also takes the residual / statistics variants, which spill) against the ring kernel on the same inputs: equality and time.  GPU box only.
    python tools/pp_check.py [dualn] [res] [stat]     # dualn: the dual-N kernel (ETAINV_DUALN) instead; which epilogue variants to include
(round 6: pp_gemm_kernel (ETAINV_PP=1) is in the EXPERIMENTS=1 library only (ETAINV_LIB=.../libetainv_hip_experiments.so); the dual-N modes run on the default library)"""
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
if "ln" in sys.argv:
    # LayerNorm consumers (row-major) and the GEGLU projection: out = rstd (x W'^T - mean s) + c on raw rows, statistics from a pass over x
    for m, c, n_out, geglu in ((524288, 320, 320, 0), (524288, 320, 960, 0), (131072, 640, 640, 0), (131072, 640, 1920, 0), (32768, 1280, 1280, 0), (32768, 1280, 3840, 0),
                               (524288, 320, 2560, 1), (131072, 640, 5120, 1), (32768, 1280, 10240, 1), (393216, 320, 2560, 1)):
        x = (torch.randn(m, c, device="cuda", generator=g) * 0.5 + 0.3).to(dt)
        w = torch.randn(n_out, c, device="cuda", generator=g) * c ** -0.5
        gamma, beta, bias = 1.0 + 0.3 * torch.randn(c, device="cuda", generator=g), 0.2 * torch.randn(c, device="cuda", generator=g), torch.randn(n_out, device="cuda", generator=g)
        wp = torch.empty(n_out, c, dtype=dt, device="cuda")
        s_vec, c_vec = torch.empty(n_out, device="cuda"), torch.empty(n_out, device="cuda")
        stat = torch.empty(m, 2, device="cuda")
        _capi.check(lib.etainv_op_ln_fold(_capi.ptr(w), _capi.ptr(gamma), _capi.ptr(beta), _capi.ptr(bias), n_out, c, geglu, 1.0, _capi.ptr(wp), _capi.ptr(s_vec), _capi.ptr(c_vec), code, st))
        _capi.check(lib.etainv_op_row_stats(_capi.ptr(x), _capi.ptr(stat), m, c, 1e-5, code, st))
        n_store = n_out // 2 if geglu else n_out
        outs, ms = {}, {}
        for pp in ("0", "1"):
            os.environ[SWITCH] = pp
            out = torch.full((m, n_store), float("nan"), dtype=dt, device="cuda")
            fn = lambda: _capi.check(lib.etainv_op_gemm_ln(_capi.ptr(x), _capi.ptr(wp), _capi.ptr(c_vec), _capi.ptr(s_vec), _capi.ptr(stat), None, _capi.ptr(out), None, None,
                                                            m, n_out, c, geglu, code, st))
            fn()
            torch.cuda.synchronize()
            outs[pp] = out
            ms[pp] = min(timeit(fn) for _ in range(3))
        ndiff = int((outs["0"] != outs["1"]).sum())
        maxd = float((outs["0"].float() - outs["1"].float()).abs().max())
        h = torch.nn.functional.layer_norm(x[:4096].float(), (c,), gamma, beta, 1e-5) @ w.t() + bias
        ref = h[:, :n_out // 2] * torch.nn.functional.gelu(h[:, n_out // 2:]) if geglu else h
        err = float((outs["1"][:4096].float() - ref).norm() / ref.norm())
        bad += not (err < 1e-2 and ndiff <= 1e-4 * out.numel())
        fl = 2.0 * m * n_out * c
        print(f"LN consumer M={m} C={c} N={n_out} geglu={geglu}: ring {ms['0']:.3f} ms {fl / ms['0'] / 1e9:7.1f} TF | dual-N {ms['1']:.3f} ms {fl / ms['1'] / 1e9:7.1f} TF | "
              f"ratio {ms['1'] / ms['0']:.3f} | elements that differ {ndiff} of {out.numel()} (max abs {maxd:.2e}) | rel err vs fp32 {err:.2e}", flush=True)
if "ablate" in sys.argv:
    # timing-only ablations of the dual-N LayerNorm consumers (ETAINV_IGEMM_DEBUG: 1 = no DMA in the loop, 2 = no epilogue): what the epilogue costs per tile
    os.environ[SWITCH] = "1"
    for m, c, n_out, geglu in ((524288, 320, 2560, 1), (131072, 640, 5120, 1), (32768, 1280, 10240, 1), (524288, 320, 960, 0), (524288, 320, 320, 0)):
        x = (torch.randn(m, c, device="cuda", generator=g) * 0.5 + 0.3).to(dt)
        wp = (torch.randn(n_out, c, device="cuda", generator=g) * c ** -0.5).to(dt)
        s_vec, c_vec = torch.randn(n_out, device="cuda", generator=g), torch.randn(n_out, device="cuda", generator=g)
        stat = torch.empty(m, 2, device="cuda")
        _capi.check(lib.etainv_op_row_stats(_capi.ptr(x), _capi.ptr(stat), m, c, 1e-5, code, st))
        out = torch.empty(m, n_out // 2 if geglu else n_out, dtype=dt, device="cuda")
        fn = lambda: _capi.check(lib.etainv_op_gemm_ln(_capi.ptr(x), _capi.ptr(wp), _capi.ptr(c_vec), _capi.ptr(s_vec), _capi.ptr(stat), None, _capi.ptr(out), None, None,
                                                        m, n_out, c, geglu, code, st))
        tiles = (m // 256) * (n_out // (256 if geglu else 320)) / float(os.environ.get("ETAINV_DUALN_GRID", "256"))
        for dbg, name in ((0, "full"), (2, "no epilogue"), (1, "no DMA"), (3, "no DMA, no epilogue"), (8, "no stores"), (16, "no GELU"), (24, "no stores, no GELU"), (64, "contiguous-KB stores")):
            os.environ["ETAINV_IGEMM_DEBUG"] = str(dbg)
            ms = min(timeit(fn) for _ in range(3))
            print(f"M={m} C={c} N={n_out} geglu={geglu} {name:22s} {ms:7.3f} ms = {ms * 1e-3 / tiles * 2.1e9:8.0f} cycles per tile at 2.1 GHz ({tiles:.0f} tiles per CU)", flush=True)
        os.environ["ETAINV_IGEMM_DEBUG"] = "0"
if "convablate" in sys.argv:
    # timing-only ablations of the ping-pong PATCH conv (an ablation build of ppconv.hip: see the note at its debug bits; ETAINV_LIB points at it)
    for rows, side, cin, cout in ((128, 64, 320, 320), (128, 32, 640, 640), (128, 32, 1280, 640), (128, 16, 1280, 1280)):
        x = (torch.randn(rows, side, side, cin, device="cuda", generator=g) * 0.5).to(dt)
        w = (torch.randn(cout, 9, cin, device="cuda", generator=g) * (9 * cin) ** -0.5).to(dt)
        bias = torch.randn(cout, device="cuda", generator=g)
        out = torch.empty(rows, side, side, cout, dtype=dt, device="cuda")
        fn = lambda: _capi.check(lib.etainv_op_conv3x3(_capi.ptr(x), None, cin, 0, _capi.ptr(w), _capi.ptr(bias), None, None, _capi.ptr(out),
                                                        rows, side, side, cout, 1, 0, 9, code, st))
        steps = (rows * side * side // 256) * (cout // 160) * (9 * cin // 64) / 256.0    # 256 x 160 x 64 K steps per CU (the dual-M kernel: two of its steps)
        for dbg, name in ((0, "full"), (2, "no epilogue"), (1 | 2, "no DMA, no epilogue"), (4 | 2, "no MFMA, no epilogue"), (8 | 2, "no fragment reads, no epilogue"),
                          (1 | 8 | 2, "MFMA + barriers only"), (4 | 8 | 2, "DMA + barriers only"), (1 | 4 | 2, "reads + barriers only")):
            os.environ["ETAINV_IGEMM_DEBUG"] = str(dbg)
            ms = min(timeit(fn) for _ in range(3))
            print(f"conv {cin}->{cout} @{side} x{rows} {name:32s} {ms:7.3f} ms = {ms * 1e-3 / steps * 2.1e9:6.0f} cycles per K step at 2.1 GHz "
                  f"(1280 MFMA cycles; {2.0 * rows * side * side * cout * 9 * cin / ms / 1e9:7.1f} TF-equivalent)", flush=True)
        os.environ["ETAINV_IGEMM_DEBUG"] = "0"
print("MISMATCHES:", bad)
