"""Per-kernel parity on a real MI355X: every HIP kernel, called through the C ABI (libetainv_hip.so), against a
plain PyTorch fp32 reference of the same op / the CPU oracle / the golden vectors.
Tolerances: relative L2 error <= 2e-3 for fp16 operands, <= 1.2e-2 for bf16 (8 significant bits), both with
fp32 accumulation; elementwise step kernels run in fp32 and must match to 1e-5."""
import ctypes as C
import json
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 2e-3, torch.bfloat16: 1.2e-2, torch.float32: 2e-5}   # (fp32: the fp32-operand kernels of csrc/f32path.hip, tests/test_fp32_gpu.py)


@pytest.fixture(scope="module")
def capi():
    from etainv import _capi
    _capi.load()
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _capi


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


# ----------------------------------------------------------------------------------------- step kernels
def test_cfg_and_ddim_step(capi):
    from oracle import schedule as sch
    lib = capi.load()
    n = 2 * 4 * 64 * 64
    u, c = rnd(n, seed=1), rnd(n, seed=2)
    out = torch.empty_like(u)
    capi.check(lib.etainv_cfg_combine(capi.ptr(u), capi.ptr(c), 7.5, capi.ptr(out), n, capi.F32, capi.stream_ptr()))
    torch.testing.assert_close(out, u + 7.5 * (c - u), rtol=1e-6, atol=1e-6)
    ac = sch.alphas_cumprod()
    for t in (0, 20, 500, 980):
        a_from, a_to = sch.ddim_inverse_coeffs(ac, t, 50)
        capi.check(lib.etainv_ddim_step(capi.ptr(u), capi.ptr(c), a_from, a_to, capi.ptr(out), n, capi.F32, capi.stream_ptr()))
        ref = sch.ddim_step(u.cpu().double(), c.cpu().double(), a_from, a_to)
        torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5)
    # empty input is a no-op
    capi.check(lib.etainv_cfg_combine(capi.ptr(u), capi.ptr(c), 7.5, capi.ptr(out), 0, capi.F32, capi.stream_ptr()))
    # fp16 io
    uh, ch = u.half(), c.half()
    oh = torch.empty_like(uh)
    capi.check(lib.etainv_cfg_combine(capi.ptr(uh), capi.ptr(ch), 7.5, capi.ptr(oh), n, capi.F16, capi.stream_ptr()))
    torch.testing.assert_close(oh.float(), uh.float() + 7.5 * (ch.float() - uh.float()), rtol=2e-3, atol=2e-3)


def _eta_step(capi, x, eps_all, g, x_prev, noise, eta, mask_map, use_mask, ac, t, S, n_img):
    from oracle import schedule as sch
    lib = capi.load()
    p = t - 1000 // S
    a_t = float(ac[t])
    a_p = float(ac[p]) if p >= 0 else float(ac[0])
    var = sch.variance(ac, t, S)
    c, hw = x.shape[1], x.shape[2] * x.shape[3]
    out_x, out_eps = torch.empty_like(x), torch.empty_like(x)
    best = torch.zeros(n_img, dtype=torch.int32, device="cuda")
    losses = torch.zeros(n_img, noise.shape[0], dtype=torch.float32, device="cuda")
    scratch = torch.zeros(n_img * 16 * 64, dtype=torch.float32, device="cuda")
    capi.check(lib.etainv_eta_backward_step(capi.ptr(x), capi.ptr(eps_all), g, capi.ptr(x_prev), capi.ptr(noise), noise.shape[0],
                                            eta, capi.ptr(mask_map) if mask_map is not None else None, 0.2, int(use_mask), a_t, a_p,
                                            var, n_img, c, hw, capi.ptr(out_x), capi.ptr(out_eps), capi.ptr(best), capi.ptr(losses),
                                            capi.ptr(scratch), capi.dtype_code(x.dtype), capi.stream_ptr()))
    return out_x, out_eps, best, losses


@pytest.mark.parametrize("name", ["lin_t980", "lin_t0", "paper_t600", "paper_t620", "paper_t980", "paper_t980_nomask"])
def test_eta_backward_step_vs_reference_golden(capi, golden, name):
    """Same seeded inputs the reference's predict_step_backward was run on (tests/golden/make_golden.py)."""
    from oracle import schedule as sch
    from tests.golden import recipes
    g = golden("eta_step")
    eta_spec, t, fp16, use_mask = recipes.ETA_CASES[name]
    inp = recipes.eta_case_inputs(name)
    assert [recipes.crc(inp[k]) for k in ("latent", "unet_out", "src_prev", "mask_map", "noise")] == list(g[f"{name}/crc"])
    eta = float(sch.eta_table(eta_spec)[t])
    x = inp["latent"].float().cuda()
    out_x, out_eps, best, losses = _eta_step(capi, x, inp["unet_out"].float().cuda(), 7.5, inp["src_prev"].float().cuda(),
                                             inp["noise"].float().reshape(10, 4, 64, 64).cuda(), eta,
                                             inp["mask_map"].float().cuda(), use_mask, sch.alphas_cumprod(), t, 50, 1)
    assert int(best[0]) == int(g[f"{name}/best"])
    gl = g[f"{name}/losses"]
    if np.isfinite(gl).all():
        np.testing.assert_allclose(losses[0].cpu().numpy(), gl, rtol=2e-4)
    np.testing.assert_allclose(out_x.cpu().numpy(), g[f"{name}/new"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("name", ["gt_thres", "fwd_t", "soft", "soft_pow", "thres_pow"])
def test_eta_backward_step_mask_modes_vs_reference_golden(capi, golden, name):
    """non-default eta-mask modes (reference get_mask, eta_inversion.py:159-205): the host prepares the per-pixel multiplier
    (threshold and / or pow) and the kernel applies it as is (use_mask = 2)"""
    from oracle import schedule as sch
    from tests.golden import recipes
    g = golden("eta_step_modes")
    mode = recipes.ETA_MODE_CASES[name]
    inp = recipes.eta_case_inputs(name)
    assert [recipes.crc(inp[k]) for k in ("latent", "unet_out", "src_prev", "mask_map", "noise")] == list(g[f"{name}/crc"])
    m = inp["mask_map"].float()
    if mode.get("thres", 0.2) is not None:
        m = (m > mode.get("thres", 0.2)).float()
    if mode.get("pow") is not None:
        m = torch.pow(m, mode["pow"])
    eta = float(sch.eta_table([[0.6, 0], [1, 0.7]])[980])
    out_x, _, _, _ = _eta_step(capi, inp["latent"].float().cuda(), inp["unet_out"].float().cuda(), 7.5, inp["src_prev"].float().cuda(),
                               inp["noise"].float().reshape(10, 4, 64, 64).cuda(), eta, m.contiguous().cuda(), 2, sch.alphas_cumprod(), 980, 50, 1)
    np.testing.assert_allclose(out_x.cpu().numpy(), g[f"{name}/new"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("name", ["tdir", "tdir_masked", "tdir_soft", "tdir_gt", "tdir_gt_eta_fwd"])
def test_eta_backward_step_target_dirinv_vs_reference_golden(capi, golden, name):
    """etainv_eta_backward_step_ex: the reference's target_dirinv / mask_dirinv options (eta_inversion.py:251-256)"""
    from oracle import schedule as sch
    from tests.golden import recipes
    g = golden("eta_step_dirinv")
    mode = recipes.ETA_DIRINV_CASES[name]
    inp = recipes.eta_case_inputs(name)
    assert [recipes.crc(inp[k]) for k in ("latent", "unet_out", "src_prev", "mask_map", "noise")] == list(g[f"{name}/crc"])
    def shaped(m):                                                    # get_mask tail (eta_inversion.py:196-201)
        m = m.float()
        if mode.get("thres", 0.2) is not None:
            m = (m > mode.get("thres", 0.2)).float()
        if mode.get("pow") is not None:
            m = torch.pow(m, mode["pow"])
        return m
    src = {"gt": recipes.dirinv_gt_mask(inp["mask_map"]), "fwd": inp["mask_map"], "fwd_mean": inp["mask_map"]}   # mask_dirinv may name another source
    m = shaped(src[mode["mask_eta"]])
    dmap = (1 - shaped(src[mode["mask_dirinv"]])).contiguous().cuda() if mode.get("mask_dirinv") else None
    lib = capi.load()
    ac, t, S = sch.alphas_cumprod(), 980, 50
    eta = float(sch.eta_table([[0.6, 0], [1, 0.7]])[t])
    x = inp["latent"].float().cuda()
    eps_all, xp, noise = inp["unet_out"].float().cuda(), inp["src_prev"].float().cuda(), inp["noise"].float().reshape(10, 4, 64, 64).cuda()
    mm = m.contiguous().cuda()
    out_x = torch.empty_like(x)
    best = torch.zeros(1, dtype=torch.int32, device="cuda")
    scratch = torch.zeros(16 * 64, dtype=torch.float32, device="cuda")
    p_ = t - 1000 // S
    capi.check(lib.etainv_eta_backward_step_ex(capi.ptr(x), capi.ptr(eps_all), 7.5, capi.ptr(xp), capi.ptr(noise), 10, eta, capi.ptr(mm), 0.0, 2,
                                               float(ac[t]), float(ac[p_]), sch.variance(ac, t, S), 1, 4, 64 * 64, capi.ptr(out_x), None, capi.ptr(best), None,
                                               capi.ptr(scratch), capi.F32, float(mode["target_dirinv"]), capi.ptr(dmap) if dmap is not None else None,
                                               capi.stream_ptr()))
    np.testing.assert_allclose(out_x.cpu().numpy(), g[f"{name}/new"], rtol=1e-4, atol=5e-5)


def test_eta_backward_step_batched_images(capi):
    """n_img = 3 pairs in the [src.., tgt..] / [u_s.., u_t.., c_s.., c_t..] layout == three B=1 calls."""
    from oracle import schedule as sch
    ac = sch.alphas_cumprod()
    B, L = 3, 32
    x = rnd(2 * B, 4, L, L, seed=3)
    eps = rnd(4 * B, 4, L, L, seed=4)
    xp = rnd(B, 4, L, L, seed=5)
    noise = rnd(10, 4, L, L, seed=6)
    mask = torch.rand(B, L, L, generator=torch.Generator().manual_seed(7)).cuda()
    ox, oe, best, _ = _eta_step(capi, x, eps, 7.5, xp, noise, 0.3, mask, True, ac, 500, 50, B)
    for i in range(B):
        xi = torch.stack([x[i], x[B + i]])
        ei = torch.stack([eps[i], eps[B + i], eps[2 * B + i], eps[3 * B + i]])
        o1, e1, b1, _ = _eta_step(capi, xi, ei, 7.5, xp[i:i + 1].contiguous(), noise, 0.3, mask[i:i + 1].contiguous(), True, ac, 500, 50, 1)
        assert int(b1[0]) == int(best[i])
        assert torch.equal(o1[0], ox[i]) and torch.equal(o1[1], ox[B + i])


# ----------------------------------------------------------------------------------------- GEMM / conv
def pack_geglu(w):
    """row interleave used by the GEGLU epilogue (csrc/misc.hip pack mode 2)."""
    rows = w.shape[0]
    p = torch.arange(rows)
    blk, within = p // 64, p % 64
    logical = torch.where(within < 32, blk * 32 + within, rows // 2 + blk * 32 + within - 32)
    return w[logical]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k", [(4096, 320, 320), (154, 640, 768), (8192, 960, 320), (64, 1280, 1280), (2, 1280, 320),
                                   (1024, 1280, 5120), (300, 2560, 1280)])
def test_gemm(capi, dtype, m, n, k):
    lib = capi.load()
    a, w = rnd(m, k, seed=1, dtype=dtype), rnd(n, k, seed=2, scale=k ** -0.5, dtype=dtype)
    bias, res = rnd(n, seed=3), rnd(m, n, seed=4, dtype=dtype)
    out = torch.empty(m, n, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_gemm(capi.ptr(a), capi.ptr(w), capi.ptr(bias), capi.ptr(res), capi.ptr(out), m, n, k, 0,
                                  capi.dtype_code(dtype), capi.stream_ptr()))
    ref = a.float() @ w.float().t() + bias + res.float()
    assert relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k,bias,res", [(16384, 960, 320, False, False), (16384, 320, 320, True, True), (65536, 320, 1280, True, True),
                                            (32768, 640, 640, True, False), (16384 + 200, 1920, 640, True, True), (65536, 160, 64, True, False)])
def test_gemm_persistent_ring(capi, dtype, m, n, k, bias, res):
    """shapes large enough for the 256 x 160 ring kernel with several tiles per persistent block (the bench configuration):
    tile switches, LDS-staged bias ring, wide stores, a ragged last M tile, nk = 1"""
    lib = capi.load()
    a, w = rnd(m, k, seed=1, dtype=dtype), rnd(n, k, seed=2, scale=k ** -0.5, dtype=dtype)
    b_ = rnd(n, seed=3) if bias else None
    r_ = rnd(m, n, seed=4, dtype=dtype) if res else None
    out = torch.empty(m, n, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_gemm(capi.ptr(a), capi.ptr(w), capi.ptr(b_), capi.ptr(r_), capi.ptr(out), m, n, k, 0,
                                  capi.dtype_code(dtype), capi.stream_ptr()))
    ref = a.float() @ w.float().t()
    if bias:
        ref = ref + b_
    if res:
        ref = ref + r_.float()
    assert relerr(out, ref) < TOL[dtype]
    # per-tile check: the worst 256-row stripe must be as good as the average (a wrong tile hides in a global norm)
    err = (out.float() - ref)[: m // 256 * 256].reshape(-1, 256, n).norm(dim=(1, 2)) / ref[: m // 256 * 256].reshape(-1, 256, n).norm(dim=(1, 2))
    assert float(err.max()) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,c", [(16384, 320), (8192, 640)])
def test_gemm_geglu_persistent_ring(capi, dtype, m, c):
    lib = capi.load()
    a, w = rnd(m, c, seed=1, dtype=dtype), rnd(8 * c, c, seed=2, scale=c ** -0.5, dtype=dtype)
    bias = rnd(8 * c, seed=3)
    out = torch.empty(m, 4 * c, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_gemm(capi.ptr(a), capi.ptr(pack_geglu(w.cpu()).cuda().contiguous()), capi.ptr(pack_geglu(bias.cpu()).cuda().contiguous()),
                                  None, capi.ptr(out), m, 8 * c, c, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    h = a.float() @ w.float().t() + bias
    ref = h[:, :4 * c] * F.gelu(h[:, 4 * c:])
    assert relerr(out, ref) < TOL[dtype]
    err = (out.float() - ref).reshape(-1, 256, 4 * c).norm(dim=(1, 2)) / ref.reshape(-1, 256, 4 * c).norm(dim=(1, 2))
    assert float(err.max()) < 2 * TOL[dtype]


def _ln_pair(capi, dtype, m, c, n_out, geglu, k_prod=None, ragged=0):
    """producer GEMM (+bias +residual) that leaves the row statistics of its stored output, then the consumer GEMM of LayerNorm(x) on the raw rows
    with gamma folded into the weights -- against x = stored rows, LayerNorm in fp32, a plain fp32 Linear"""
    lib = capi.load()
    dt = capi.dtype_code(dtype)
    k_prod = k_prod or c
    m = m + ragged
    a0, w0 = rnd(m, k_prod, seed=1, dtype=dtype), rnd(c, k_prod, seed=2, scale=k_prod ** -0.5, dtype=dtype)
    b0 = rnd(c, seed=3) + 0.7                      # a row mean well away from zero
    r0 = rnd(m, c, seed=4, scale=2.0, dtype=dtype)
    x = torch.empty(m, c, dtype=dtype, device="cuda")
    part = torch.full((m * (c // 32) * 2,), float("nan"), dtype=torch.float32, device="cuda")
    stat = torch.full((m, 2), float("nan"), dtype=torch.float32, device="cuda")
    sp = C.c_int(-1)
    capi.check(lib.etainv_op_gemm_ln(capi.ptr(a0), capi.ptr(w0), capi.ptr(b0), None, None, capi.ptr(r0), capi.ptr(x), capi.ptr(part), C.byref(sp),
                                     m, c, k_prod, 0, dt, capi.stream_ptr()))
    P = sp.value
    xr = a0.float() @ w0.float().t() + b0 + r0.float()
    assert relerr(x, xr) < TOL[dtype]
    xf = x.float()                                 # statistics are those of the STORED (rounded) rows
    if P == 0:                                     # this launch shape cannot emit partials: the pass the engine falls back to
        capi.check(lib.etainv_op_row_stats(capi.ptr(x), capi.ptr(stat), m, c, 1e-5, dt, capi.stream_ptr()))
    else:
        st = part[: m * P * 2].view(m, P, 2)
        assert torch.isfinite(st).all()
        xs = xf.view(m, P, c // P)
        torch.testing.assert_close(st[..., 0], xs.mean(-1), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(st[..., 1], ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1), rtol=2e-4, atol=1e-3)
        capi.check(lib.etainv_op_ln_finalize(capi.ptr(part), P, c // P, 1e-5, capi.ptr(stat), m, capi.stream_ptr()))
    torch.testing.assert_close(stat[:, 0], xf.mean(-1), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(stat[:, 1], (xf.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=1e-6)
    # consumer
    w = rnd(n_out, c, seed=5, scale=c ** -0.5)
    gamma, beta, bias = 1.0 + 0.3 * rnd(c, seed=6), 0.2 * rnd(c, seed=7), rnd(n_out, seed=8)
    wp = torch.empty(n_out, c, dtype=dtype, device="cuda")
    s_vec, c_vec = torch.empty(n_out, device="cuda"), torch.empty(n_out, device="cuda")
    capi.check(lib.etainv_op_ln_fold(capi.ptr(w), capi.ptr(gamma), capi.ptr(beta), capi.ptr(bias), n_out, c, geglu, 1.0, capi.ptr(wp), capi.ptr(s_vec),
                                     capi.ptr(c_vec), dt, capi.stream_ptr()))
    n_store = n_out // 2 if geglu else n_out
    out = torch.empty(m, n_store, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_gemm_ln(capi.ptr(x), capi.ptr(wp), capi.ptr(c_vec), capi.ptr(s_vec), capi.ptr(stat), None, capi.ptr(out), None, None,
                                     m, n_out, c, geglu, dt, capi.stream_ptr()))
    h = F.layer_norm(xf, (c,), gamma, beta, 1e-5) @ w.t() + bias
    ref = h[:, :n_out // 2] * F.gelu(h[:, n_out // 2:]) if geglu else h
    assert relerr(out, ref) < TOL[dtype]
    rows = m // 64 * 64
    if rows:
        err = (out.float() - ref)[:rows].reshape(-1, 64, n_store).norm(dim=(1, 2)) / ref[:rows].reshape(-1, 64, n_store).norm(dim=(1, 2))
        assert float(err.max()) < 2 * TOL[dtype]   # a wrong row group hides in a global norm
    return sp.value


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,c,n_out,geglu,expect_p", [
    (32768, 320, 960, 0, 4),        # 256 x 160 ring on both sides (fused QKV of the 64 x 64 level): 4 partials of 80 columns
    (16384, 640, 640, 0, 8),        # attn2.to_q
    (32768, 320, 2560, 1, 4),       # GEGLU projection on the 256 x 128 ring
    (4096, 320, 960, 0, None),      # mid-size tiles (the producer runs 64 x 64 tiles here: 10 partials)
    (8192, 1280, 1280, 0, 16),      # 16 partials per row
    (1024, 320, 2560, 1, None),     # GEGLU on 128 x 128 tiles
    (256, 320, 320, 0, None),       # 64 x 64 tiles
    (64, 1280, 3840, 0, None),      # producer AND consumer split along K: statistics by the fallback pass, LayerNorm in the reduction kernel
])
def test_gemm_layernorm_fold(capi, dtype, m, c, n_out, geglu, expect_p):
    p = _ln_pair(capi, dtype, m, c, n_out, geglu)
    if expect_p is not None:
        assert p == expect_p


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m_tiles,n_out,geglu", [(256, 2560, 1), (700, 2560, 1), (256, 960, 0), (513, 960, 0), (1024, 1920, 0)])
def test_xsgemm_equals_ring_kernel(capi, dtype, monkeypatch, m_tiles, n_out, geglu):
    """csrc/xsgemm.hip (opt-in experiment, ETAINV_XSGEMM=1: K = 320 LayerNorm consumers with many rows on a stationary activation tile, two wave
    groups in anti-phase) accumulates in the ring kernel's K order and runs its epilogue arithmetic: the two must give EQUAL values.  Row counts that give the 256 persistent blocks 1, 2-3
    (uneven) and 4 M tiles each; GEGLU (128-column N tiles) and plain (96-column) epilogues."""
    import ctypes
    try:    # an opt-in experiment that lost: `EXPERIMENTS=1 bash csrc/build.sh` + ETAINV_LIB=.../libetainv_hip_experiments.so (round 6: out of the default library)
        ctypes.c_int.in_dll(capi.load(), "etainv_experiments_built")
    except ValueError:
        pytest.skip("xsgemm.hip is not in the default library (EXPERIMENTS=1 build only)")
    lib = capi.load()
    dt = capi.dtype_code(dtype)
    m, c = 128 * m_tiles, 320
    x = rnd(m, c, seed=11, scale=1.5, dtype=dtype) + 0.3
    xf = x.float()
    stat = torch.stack([xf.mean(-1), (xf.var(-1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    w = rnd(n_out, c, seed=5, scale=c ** -0.5)
    gamma, beta, bias = 1.0 + 0.3 * rnd(c, seed=6), 0.2 * rnd(c, seed=7), rnd(n_out, seed=8)
    wp = torch.empty(n_out, c, dtype=dtype, device="cuda")
    s_vec, c_vec = torch.empty(n_out, device="cuda"), torch.empty(n_out, device="cuda")
    capi.check(lib.etainv_op_ln_fold(capi.ptr(w), capi.ptr(gamma), capi.ptr(beta), capi.ptr(bias), n_out, c, geglu, 1.0, capi.ptr(wp), capi.ptr(s_vec),
                                     capi.ptr(c_vec), dt, capi.stream_ptr()))
    n_store = n_out // 2 if geglu else n_out
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("ETAINV_XSGEMM", on)
        out = torch.full((m, n_store), float("nan"), dtype=dtype, device="cuda")
        capi.check(lib.etainv_op_gemm_ln(capi.ptr(x), capi.ptr(wp), capi.ptr(c_vec), capi.ptr(s_vec), capi.ptr(stat), None, capi.ptr(out), None, None,
                                         m, n_out, c, geglu, dt, capi.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
    h = F.layer_norm(xf, (c,), gamma, beta, 1e-5) @ w.t() + bias
    ref = h[:, :n_out // 2] * F.gelu(h[:, n_out // 2:]) if geglu else h
    assert relerr(outs[0], ref) < TOL[dtype] and relerr(outs[1], ref) < TOL[dtype]
    # same MFMA accumulation order, same epilogue expressions -- but the compiler contracts a * b + c into an fma where it sees fit in each kernel:
    # measured 619 of 41.9 M results land on the other side of a rounding boundary (one ulp of the 16-bit output; a zero may change sign)
    a, b = outs[0].float(), outs[1].float()
    neq = a != b
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    worst = float(((a - b).abs() / b.abs().clamp_min(1e-30))[neq].max()) if bool(neq.any()) else 0.0
    print(f"xsgemm vs ring: {int(neq.sum())} of {neq.numel()} elements differ, worst relative difference {worst:.3e}")
    assert int(neq.sum()) <= 1e-4 * neq.numel() and worst <= 1.01 * ulp, f"rows {neq.any(1).nonzero().flatten()[:8].tolist()}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_layernorm_fold_ragged(capi, dtype):
    """row counts that are not a multiple of the wave tile: general epilogue on the consumer, fallback statistics pass on the producer"""
    assert _ln_pair(capi, dtype, 4096, 320, 960, 0, ragged=24) == 0
    assert _ln_pair(capi, dtype, 32768, 320, 960, 0, ragged=40) == 0
    _ln_pair(capi, dtype, 100, 640, 5120, 1, ragged=0)
    _ln_pair(capi, dtype, 12, 1280, 1280, 0, ragged=0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_persistent_ring(capi, dtype):
    """4 x 64 x 64 x 320 -> 320 (M = 16384, K = 2880, 128 tiles of 256 x 160) and 8 x 32 x 32 x 1280+640 -> 640"""
    lib = capi.load()
    for b, h, c1, c2, cout, ups, stride in ((8, 64, 320, 0, 320, 0, 1), (8, 32, 1280, 640, 640, 0, 1), (8, 32, 640, 0, 640, 1, 1),
                                            (16, 64, 320, 0, 320, 0, 2)):   # + fused 2x upsample and stride 2 at bench-sized M
        cin = c1 + c2
        x = rnd(b, cin, h, h, seed=1, dtype=dtype)
        w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5, dtype=dtype)
        bias, rowvec = rnd(cout, seed=3), rnd(b, cout, seed=5)
        xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
        ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1) + rowvec[:, :, None, None]
        ho = ref.shape[-1]
        x_nhwc = x.permute(0, 2, 3, 1).contiguous()
        x1 = x_nhwc[..., :c1].contiguous()
        x2 = x_nhwc[..., c1:].contiguous() if c2 else None
        out = torch.empty(b, ho, ho, cout, dtype=dtype, device="cuda")
        capi.check(lib.etainv_op_conv3x3(capi.ptr(x1), capi.ptr(x2), c1, c2, capi.ptr(w.permute(0, 2, 3, 1).contiguous()), capi.ptr(bias), capi.ptr(rowvec),
                                         None, capi.ptr(out), b, h, h, cout, stride, ups, 9, capi.dtype_code(dtype), capi.stream_ptr()))
        assert relerr(out.permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,h,wd,cin,cout", [(16, 16, 16, 1280, 1280), (8, 32, 32, 640, 640), (12, 16, 32, 320, 640),
                                             (64, 8, 8, 1280, 1280), (16, 24, 24, 640, 640), (64, 8, 16, 640, 320)])
def test_conv3x3_fused_upsample_phase_form(capi, dtype, b, h, wd, cin, cout):
    """conv3x3(nearest-2x upsample(x)) as four 2 x 2 phase convs on the source image (etainv_op_pack_ups4 + upsample = 2, taps = 4: 4 / 9 of the FLOPs):
    against F.conv2d on the upsampled image (borders of the upsampled grid = zero padding; every phase; non-square images) and against the 9-tap fused
    form (upsample = 1) -- the two differ only by the one rounding of the summed weights.  Source images of whole 256-row tiles run image-major virtual
    rows; 8 x 8, 24 x 24, 8 x 16 sources run PHASE-major rows (a tile inside one phase, its rows in several images)"""
    lib = capi.load()
    x = rnd(b, cin, h, wd, seed=1, dtype=dtype)
    w32 = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5)
    bias = rnd(cout, seed=3)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w32, bias, padding=1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    code = capi.dtype_code(dtype)
    w4 = torch.empty(4, cout, 4, cin, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_pack_ups4(capi.ptr(w32.contiguous()), capi.ptr(w4), cout, cin, code, capi.stream_ptr()))
    # the packed phase kernels are the fp32 sums of the taps that share a source pixel
    py0 = torch.stack([w32[:, :, 0], w32[:, :, 1] + w32[:, :, 2]], 2)            # [cout][cin][ty][kx], py = 0
    want00 = torch.stack([py0[..., 0], py0[..., 1] + py0[..., 2]], 3)            # px = 0 -> [cout][cin][ty][tx]
    assert relerr(w4[0].float().reshape(cout, 2, 2, cin).permute(0, 3, 1, 2), want00) < (1e-3 if dtype == torch.float16 else 5e-3)
    out = torch.empty(b, 2 * h, 2 * wd, cout, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_conv3x3(capi.ptr(x_nhwc), None, cin, 0, capi.ptr(w4), capi.ptr(bias), None, None, capi.ptr(out), b, h, wd, cout, 1, 2, 4, code,
                                     capi.stream_ptr()))
    out9 = torch.empty_like(out)
    w9 = w32.to(dtype).permute(0, 2, 3, 1).contiguous()
    capi.check(lib.etainv_op_conv3x3(capi.ptr(x_nhwc), None, cin, 0, capi.ptr(w9), capi.ptr(bias), None, None, capi.ptr(out9), b, h, wd, cout, 1, 1, 9, code,
                                     capi.stream_ptr()))
    e4, e9 = relerr(out.permute(0, 3, 1, 2), ref), relerr(out9.permute(0, 3, 1, 2), ref)
    print(f"phase form {e4:.2e}, 9-tap form {e9:.2e} vs fp32")
    assert e4 < TOL[dtype] and e9 < TOL[dtype] and e4 < 1.2 * e9 + 1e-4
    # per (image, output row) worst case: a misplaced phase row hides in a global norm
    d = (out.float() - ref.permute(0, 2, 3, 1)).reshape(b * 2 * h, -1).norm(dim=1) / ref.permute(0, 2, 3, 1).reshape(b * 2 * h, -1).norm(dim=1)
    assert float(d.max()) < 3 * TOL[dtype]
    # no launch for what the ring cannot do: a source image whose 64-row wave tiles would straddle images (5 x 5 = 25 pixels)
    with pytest.raises(Exception, match="phase form"):
        capi.check(lib.etainv_op_conv3x3(capi.ptr(x_nhwc), None, cin, 0, capi.ptr(w4), capi.ptr(bias), None, None, capi.ptr(out), b, 5, 5, cout, 1, 2, 4, code,
                                         capi.stream_ptr()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,h,wd,cin,cout,res", [(16, 64, 64, 320, 320, True), (16, 64, 64, 960, 320, False), (64, 32, 32, 640, 640, True),
                                                 (256, 16, 16, 1280, 1280, False), (7, 48, 48, 320, 640, True), (48, 16, 32, 640, 640, True),
                                                 (96, 32, 16, 320, 320, False)])
def test_conv3x3_patch_mode(capi, dtype, monkeypatch, b, h, wd, cin, cout, res):
    """igemm.hip PATCH mode (ETAINV_PATCHCONV=1): an M tile is a 16 x 16 pixel patch whose halo'd 18 x 18 activation patch is brought to LDS once per
    channel chunk for all nine taps (K loop chunk-major).  Against F.conv2d and against the tap-major ring kernel (same fp32 accumulation of the same
    products in another order: equal up to summation order); image borders, tile borders inside an image, non-square images (the dispatch takes any H, W that are multiples of 16), bias + time row + residual epilogues."""
    lib = capi.load()
    x = rnd(b, cin, h, wd, seed=1, dtype=dtype)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5, dtype=dtype)
    bias, rowvec = rnd(cout, seed=3), rnd(b, cout, seed=5)
    r = rnd(b, h, wd, cout, seed=6, dtype=dtype) if res else None
    ref = F.conv2d(x.float(), w.float(), bias, padding=1) + rowvec[:, :, None, None]
    if res:
        ref = ref + r.float().permute(0, 3, 1, 2)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    wk = w.permute(0, 2, 3, 1).contiguous()
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("ETAINV_PATCHCONV", on)
        out = torch.full((b, h, wd, cout), float("nan"), dtype=dtype, device="cuda")
        capi.check(lib.etainv_op_conv3x3(capi.ptr(x_nhwc), None, cin, 0, capi.ptr(wk), capi.ptr(bias), capi.ptr(rowvec), capi.ptr(r), capi.ptr(out), b, h, wd, cout,
                                         1, 0, 9, capi.dtype_code(dtype), capi.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
    e1, e0, ed = relerr(outs[0].permute(0, 3, 1, 2), ref), relerr(outs[1].permute(0, 3, 1, 2), ref), relerr(outs[0], outs[1])
    print(f"conv3x3 {b}x{h}x{wd}x{cin}->{cout} {dtype}: patch mode {e1:.2e}, ring {e0:.2e} vs F.conv2d; patch vs ring {ed:.2e}")
    assert e1 < TOL[dtype] and e0 < TOL[dtype]
    assert ed < 0.5 * TOL[dtype]
    # per-image, per-patch-row worst case (a misplaced row hides in a global norm)
    d = (outs[0].float() - ref.permute(0, 2, 3, 1)).reshape(b * h, -1).norm(dim=1) / ref.permute(0, 2, 3, 1).reshape(b * h, -1).norm(dim=1)
    assert float(d.max()) < 3 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,hw,n,k,res", [(16, 4096, 320, 320, True), (48, 1024, 640, 640, True), (192, 256, 1280, 1280, False)])
def test_gemm_dual_n_groupnorm_statistics(capi, dtype, monkeypatch, b, hw, n, k, res):
    """the dual-N kernel as a GroupNorm producer (proj_out of the transformer blocks: 1x1 conv + residual whose output feeds the next ResnetBlock's
    GroupNorm): per-channel sum / sum of squares of the stored output per 64-row block, against a pass over the output and, bit for bit, the ring kernel"""
    lib = capi.load()
    dt = capi.dtype_code(dtype)
    m = b * hw
    a, w = rnd(m, k, seed=1, dtype=dtype), rnd(n, k, seed=2, scale=k ** -0.5, dtype=dtype)
    bias = rnd(n, seed=3) + 0.3
    r_ = rnd(m, n, seed=4, dtype=dtype) if res else None
    outs, parts = [], []
    for on in ("1", "0"):
        monkeypatch.setenv("ETAINV_DUALN", on)
        out = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
        part = torch.full((m // 16 * 2 * n,), float("nan"), device="cuda")
        wm = C.c_int(-1)
        capi.check(lib.etainv_op_gemm_gnstat(capi.ptr(a), capi.ptr(w), capi.ptr(bias), capi.ptr(r_), capi.ptr(out), capi.ptr(part), C.byref(wm), m, n, k, hw,
                                             dt, capi.stream_ptr()))
        torch.cuda.synchronize()
        assert wm.value == 64
        outs.append(out)
        parts.append(part[: m // 64 * 2 * n].clone())
    ref = a.float() @ w.float().t() + bias + (r_.float() if res else 0)
    assert relerr(outs[0], ref) < TOL[dtype]
    assert torch.equal(outs[0], outs[1])
    st = parts[0].view(m // 64, 2, n)
    xf = outs[0].float().view(m // 64, 64, n)
    torch.testing.assert_close(st[:, 0], xf.sum(1), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(st[:, 1], (xf * xf).sum(1), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(parts[0], parts[1], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,h,wd,cin,cout,res", [(16, 64, 64, 320, 320, True), (16, 64, 64, 960, 320, False), (64, 32, 32, 640, 640, True),
                                                 (256, 16, 16, 1280, 1280, False), (48, 16, 32, 640, 640, True), (7, 48, 48, 320, 640, True)])
def test_conv3x3_ping_pong_patch(capi, dtype, monkeypatch, b, h, wd, cin, cout, res):
    """ppconv.hip (default for conv3x3 stride 1 on 16-pixel-aligned images): the PATCH-mode K loop with the wave groups in anti-phase and a lean issue side.
    Three kernels on the same input: the dual-M form (two 16 x 16 patches per tile, 32-channel chunks; default where the patch count is even and the last
    round of tiles is full enough), the 256-pixel form (ETAINV_PPCONV2=0) and igemm.hip's PATCH ring (ETAINV_PPCONV=0).  Against F.conv2d; the 256-pixel
    form bit for bit against the ring (same accumulation order), the dual-M form within a rounding of it (K order (32-channel chunk, tap)); the GroupNorm
    partials of the stored output (per channel sum / sum of squares over every 64-row block) against a pass over that output; image borders, several
    tiles per block, an odd tile count, non-square"""
    lib = capi.load()
    dt = capi.dtype_code(dtype)
    x = rnd(b, cin, h, wd, seed=1, dtype=dtype)
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5, dtype=dtype)
    bias, rowvec = rnd(cout, seed=3), rnd(b, cout, seed=5)
    r = rnd(b, h, wd, cout, seed=6, dtype=dtype) if res else None
    ref = F.conv2d(x.float(), w.float(), bias, padding=1) + rowvec[:, :, None, None]
    if res:
        ref = ref + r.float().permute(0, 3, 1, 2)
    x_nhwc, wk = x.permute(0, 2, 3, 1).contiguous(), w.permute(0, 2, 3, 1).contiguous()
    outs, parts, wms = [], [], []
    for pp, pp2 in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("ETAINV_PPCONV", pp)
        monkeypatch.setenv("ETAINV_PPCONV2", pp2)
        out = torch.full((b, h, wd, cout), float("nan"), dtype=dtype, device="cuda")
        part = torch.full((b * h * wd // 16 * 2 * cout,), float("nan"), device="cuda")
        wm = C.c_int(-1)
        capi.check(lib.etainv_op_conv3x3_gnstat(capi.ptr(x_nhwc), capi.ptr(wk), capi.ptr(bias), capi.ptr(rowvec), capi.ptr(r), capi.ptr(out), capi.ptr(part),
                                                C.byref(wm), b, h, wd, cin, cout, dt, capi.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
        parts.append(part)
        wms.append(wm.value)
        out2 = torch.full((b, h, wd, cout), float("nan"), dtype=dtype, device="cuda")        # the plain epilogue (no statistics)
        capi.check(lib.etainv_op_conv3x3(capi.ptr(x_nhwc), None, cin, 0, capi.ptr(wk), capi.ptr(bias), capi.ptr(rowvec), capi.ptr(r), capi.ptr(out2), b, h, wd, cout,
                                         1, 0, 9, dt, capi.stream_ptr()))
        assert torch.equal(out, out2)
    n_blk = b * h * wd // 64
    for o, pt in zip(outs[:2], parts[:2]):
        assert relerr(o.permute(0, 3, 1, 2), ref) < TOL[dtype]
        d = (o.float() - ref.permute(0, 2, 3, 1)).reshape(b * h, -1).norm(dim=1) / ref.permute(0, 2, 3, 1).reshape(b * h, -1).norm(dim=1)
        assert float(d.max()) < 3 * TOL[dtype]             # per image row: a misplaced patch row hides in a global norm
        # partials: blocks of 64 VIRTUAL rows (patches of 16 x 16 pixels enumerated image-major, 4 patch rows per block) -> compare per image
        st = pt[: n_blk * 2 * cout].view(b, h * wd // 64, 2, cout).sum(1)
        xf = o.float().reshape(b, h * wd, cout)
        torch.testing.assert_close(st[:, 0], xf.sum(1), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(st[:, 1], (xf * xf).sum(1), rtol=1e-4, atol=1e-2)
    assert torch.equal(outs[1], outs[2]), "the 256-pixel ping-pong kernel and the ring PATCH kernel accumulate in the same order"
    torch.testing.assert_close(parts[1][: n_blk * 2 * cout], parts[2][: n_blk * 2 * cout], rtol=1e-6, atol=1e-6)
    assert wms[0] == wms[1] == wms[2] == 64
    # dual-M vs the ring: the same products summed in another order -- a rounding of the stored type at most, in a small share of the elements
    differ = outs[0] != outs[2]
    assert float(differ.float().mean()) < 0.12
    assert relerr(outs[0], outs[2]) < 0.25 * TOL[dtype]
    one_ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    assert float(((outs[0].float() - outs[2].float()).abs() / outs[2].float().abs().clamp_min(0.25)).max()) <= 1.01 * one_ulp


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,side,c1,c2,n", [(16, 64, 320, 320, 320), (16, 64, 640, 320, 320), (64, 32, 640, 640, 640), (48, 32, 1280, 640, 640), (256, 16, 1280, 1280, 1280)])
def test_gemm_dual_n_two_sources(capi, dtype, monkeypatch, b, side, c1, c2, n):
    """the 1x1 shortcut of an up-block resnet over [hidden | skip] without materialising the concatenation (reference: diffusers ResnetBlock2D.conv_shortcut
    inside the UNet call of eta_inversion.py:321): dual-N kernel with two activation sources (K tiles 0 .. c1 / 64 - 1 from the first, the rest from the
    second) against torch on the concatenated input and, bit for bit, against the ring kernel (ETAINV_DUALN_A2=0: same K order)"""
    lib = capi.load()
    dt = capi.dtype_code(dtype)
    x1, x2 = rnd(b, side, side, c1, seed=1, dtype=dtype), rnd(b, side, side, c2, seed=2, dtype=dtype)
    w = rnd(n, 1, c1 + c2, seed=3, scale=(c1 + c2) ** -0.5, dtype=dtype)
    bias = rnd(n, seed=4)
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("ETAINV_DUALN_A2", on)
        out = torch.full((b, side, side, n), float("nan"), dtype=dtype, device="cuda")
        capi.check(lib.etainv_op_conv3x3(capi.ptr(x1), capi.ptr(x2), c1, c2, capi.ptr(w), capi.ptr(bias), None, None, capi.ptr(out), b, side, side, n, 1, 0, 1, dt,
                                         capi.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
    ref = torch.cat([x1, x2], -1).reshape(-1, c1 + c2)[:8192].float() @ w.reshape(n, -1).float().t() + bias
    assert relerr(outs[0].reshape(-1, n)[:8192], ref) < TOL[dtype]
    assert torch.equal(outs[0], outs[1]), "dual-N and ring accumulate the K tiles of the two sources in the same order"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,k,res,stat", [(65536, 320, 320, True, True), (65536, 320, 1280, True, True), (49152, 640, 640, True, False),
                                            (49152, 640, 2560, False, True), (49152, 1280, 5120, True, True), (98304, 320, 128, False, False),
                                            (131072 + 256 * 3, 320, 320, True, True)])
def test_gemm_dual_n_ping_pong(capi, dtype, monkeypatch, m, n, k, res, stat):
    """ppgemm.hip pp_dualn_kernel (default for 1x1 / Linear launches on whole 256 x 320 tiles with bias / residual / LayerNorm-statistics epilogues): two
    160-column halves from one staged activation K tile, wave groups in anti-phase, 2-slot LDS ring recycled by region.  Against fp32 and, bit for bit,
    against the ring kernel (same accumulation order); several tiles per block, an odd tile count per block, K tiles 2 .. 80, the row statistics"""
    lib = capi.load()
    dt = capi.dtype_code(dtype)
    a, w = rnd(m, k, seed=1, dtype=dtype), rnd(n, k, seed=2, scale=k ** -0.5, dtype=dtype)
    b_ = rnd(n, seed=3) + 0.5
    r_ = rnd(m, n, seed=4, scale=2.0, dtype=dtype) if res else None
    outs, parts, ps = [], [], []
    for on in ("1", "0"):
        monkeypatch.setenv("ETAINV_DUALN", on)
        out = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
        part = torch.full((m * (n // 32) * 2,), float("nan"), dtype=torch.float32, device="cuda") if stat else None
        sp = C.c_int(-1)
        capi.check(lib.etainv_op_gemm_ln(capi.ptr(a), capi.ptr(w), capi.ptr(b_), None, None, capi.ptr(r_), capi.ptr(out), capi.ptr(part),
                                         C.byref(sp) if stat else None, m, n, k, 0, dt, capi.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
        parts.append(part)
        ps.append(sp.value)
    ref = a.float() @ w.float().t() + b_
    if res:
        ref = ref + r_.float()
    assert relerr(outs[0], ref) < TOL[dtype]
    err = (outs[0].float() - ref).reshape(-1, 256, n).norm(dim=(1, 2)) / ref.reshape(-1, 256, n).norm(dim=(1, 2))
    assert float(err.max()) < 2 * TOL[dtype]          # a wrong tile hides in a global norm
    assert torch.equal(outs[0], outs[1]), "dual-N kernel and ring kernel accumulate in the same order"
    if stat:
        assert ps[0] == ps[1] == n // 80
        P = ps[0]
        st = parts[0][: m * P * 2].view(m, P, 2)
        xs = outs[0].float().view(m, P, n // P)
        torch.testing.assert_close(st[..., 0], xs.mean(-1), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(st[..., 1], ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1), rtol=2e-4, atol=1e-3)
        torch.testing.assert_close(parts[0][: m * P * 2], parts[1][: m * P * 2], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
def test_split_k_small_m_deep_k(capi, dtype):
    """batch-1 shapes of the 8x8 / 16x16 levels (M = 64 .. 256, K = 11520 / 23040): split-K partials + fixed-order reduction with
    the fused bias / time-embedding row / residual epilogue; a plain GEMM with M = 64, K = 5120 as well"""
    lib = capi.load()
    for b, h, c1, c2, cout in ((1, 8, 1280, 0, 1280), (4, 8, 1280, 1280, 1280), (1, 16, 1280, 0, 1280)):
        cin = c1 + c2
        x = rnd(b, cin, h, h, seed=1, dtype=dtype)
        w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5, dtype=dtype)
        bias, rowvec = rnd(cout, seed=3), rnd(b, cout, seed=5)
        res = rnd(b, h, h, cout, seed=6, dtype=dtype)
        ref = F.conv2d(x.float(), w.float(), bias, padding=1) + rowvec[:, :, None, None] + res.float().permute(0, 3, 1, 2)
        x_nhwc = x.permute(0, 2, 3, 1).contiguous()
        x1 = x_nhwc[..., :c1].contiguous()
        x2 = x_nhwc[..., c1:].contiguous() if c2 else None
        out = torch.empty(b, h, h, cout, dtype=dtype, device="cuda")
        capi.check(lib.etainv_op_conv3x3(capi.ptr(x1), capi.ptr(x2), c1, c2, capi.ptr(w.permute(0, 2, 3, 1).contiguous()), capi.ptr(bias), capi.ptr(rowvec),
                                         capi.ptr(res), capi.ptr(out), b, h, h, cout, 1, 0, 9, capi.dtype_code(dtype), capi.stream_ptr()))
        assert relerr(out.permute(0, 3, 1, 2), ref) < TOL[dtype]
    m, n, k = 64, 1280, 5120
    a, w = rnd(m, k, seed=1, dtype=dtype), rnd(n, k, seed=2, scale=k ** -0.5, dtype=dtype)
    bias, res = rnd(n, seed=3), rnd(m, n, seed=4, dtype=dtype)
    out = torch.empty(m, n, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_gemm(capi.ptr(a), capi.ptr(w), capi.ptr(bias), capi.ptr(res), capi.ptr(out), m, n, k, 0, capi.dtype_code(dtype), capi.stream_ptr()))
    assert relerr(out, a.float() @ w.float().t() + bias + res.float()) < TOL[dtype]
    out2 = torch.empty_like(out)
    capi.check(lib.etainv_op_gemm(capi.ptr(a), capi.ptr(w), capi.ptr(bias), capi.ptr(res), capi.ptr(out2), m, n, k, 0, capi.dtype_code(dtype), capi.stream_ptr()))
    assert torch.equal(out, out2)      # deterministic


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_geglu(capi, dtype):
    lib = capi.load()
    m, c = 1024, 320
    a, w = rnd(m, c, seed=1, dtype=dtype), rnd(8 * c, c, seed=2, scale=c ** -0.5, dtype=dtype)
    bias = rnd(8 * c, seed=3)
    out = torch.empty(m, 4 * c, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_gemm(capi.ptr(a), capi.ptr(pack_geglu(w.cpu()).cuda().contiguous()), capi.ptr(pack_geglu(bias.cpu()).cuda().contiguous()),
                                  None, capi.ptr(out), m, 8 * c, c, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    h = a.float() @ w.float().t() + bias
    ref = h[:, :4 * c] * F.gelu(h[:, 4 * c:])
    assert relerr(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [dict(b=2, h=32, c1=320, c2=0, cout=320, stride=1, ups=0),
                                 dict(b=2, h=16, c1=640, c2=0, cout=640, stride=2, ups=0),
                                 dict(b=1, h=16, c1=640, c2=0, cout=640, stride=1, ups=1),
                                 dict(b=2, h=16, c1=640, c2=320, cout=320, stride=1, ups=0),
                                 dict(b=3, h=8, c1=1280, c2=1280, cout=1280, stride=1, ups=0),
                                 dict(b=1, h=12, c1=320, c2=0, cout=64, stride=1, ups=0)])
def test_conv3x3(capi, dtype, cfg):
    lib = capi.load()
    b, h, c1, c2, cout = cfg["b"], cfg["h"], cfg["c1"], cfg["c2"], cfg["cout"]
    cin = c1 + c2
    x = rnd(b, cin, h, h, seed=1, dtype=dtype)                              # NCHW reference input
    w = rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5, dtype=dtype)
    bias = rnd(cout, seed=3)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if cfg["ups"] else x.float()
    ref = F.conv2d(xin, w.float(), bias, stride=cfg["stride"], padding=1)
    ho = ref.shape[-1]
    rowvec = rnd(b, cout, seed=5)
    res = rnd(b, ho, ho, cout, seed=6, dtype=dtype)
    ref = ref + rowvec[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    x1 = x_nhwc[..., :c1].contiguous()
    x2 = x_nhwc[..., c1:].contiguous() if c2 else None
    w_p = w.permute(0, 2, 3, 1).contiguous()                                 # [O][ky][kx][I]
    out = torch.empty(b, ho, ho, cout, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_conv3x3(capi.ptr(x1), capi.ptr(x2), c1, c2, capi.ptr(w_p), capi.ptr(bias), capi.ptr(rowvec), capi.ptr(res),
                                     capi.ptr(out), b, h, h, cout, cfg["stride"], cfg["ups"], 9, capi.dtype_code(dtype), capi.stream_ptr()))
    assert relerr(out.permute(0, 3, 1, 2), ref) < TOL[dtype]


# ----------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,hw,c1,c2,silu", [(2, 1024, 320, 0, 1), (3, 256, 1280, 640, 1), (1, 4096, 640, 320, 1), (2, 64, 1280, 1280, 1),
                                             (2, 256, 640, 0, 0), (1, 144, 320, 0, 1),
                                             # one-pass kernel (chunks of an image held in registers, partial sums handed over inside the launch):
                                             # several images per block group, dual source, ragged last chunk, groups that straddle a vector
                                             (9, 4096, 320, 0, 1), (40, 1024, 320, 0, 0), (5, 1024, 320, 320, 1), (3, 200, 256, 64, 1), (12, 256, 640, 0, 1)])
def test_groupnorm(capi, dtype, b, hw, c1, c2, silu):
    lib = capi.load()
    C_ = c1 + c2
    x = (rnd(b, hw, C_, seed=1) * 1.5 + 0.3).to(dtype)
    gamma, beta = rnd(C_, seed=2) * 0.1 + 1, rnd(C_, seed=3) * 0.1
    x1 = x[..., :c1].contiguous()
    x2 = x[..., c1:].contiguous() if c2 else None
    out = torch.empty(b, hw, C_, dtype=dtype, device="cuda")
    scratch = torch.zeros(b * 65 * 32 * 2, dtype=torch.float32, device="cuda")
    eps = 1e-5 if silu else 1e-6
    capi.check(lib.etainv_op_groupnorm(capi.ptr(x1), capi.ptr(x2), c1, c2, capi.ptr(gamma), capi.ptr(beta), capi.ptr(out), b, hw, 32, eps,
                                       silu, capi.ptr(scratch), capi.dtype_code(dtype), capi.stream_ptr()))
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    assert relerr(out, ref.permute(0, 2, 1)) < TOL[dtype]


def _gemm_gnstat(capi, dtype, b, hw, c, k, seed, residual):
    """x = a W^T + bias (+ residual) through the GEMM whose epilogue leaves the per-channel GroupNorm partials of the stored x"""
    lib = capi.load()
    m = b * hw
    a, w = rnd(m, k, seed=seed, dtype=dtype), rnd(c, k, seed=seed + 1, scale=k ** -0.5, dtype=dtype)
    bias = rnd(c, seed=seed + 2) + 0.4
    res = rnd(m, c, seed=seed + 3, dtype=dtype) if residual else None
    x = torch.empty(m, c, dtype=dtype, device="cuda")
    part = torch.full((m // 32 * 2 * c + 64,), float("nan"), dtype=torch.float32, device="cuda")
    wm = C.c_int(-1)
    capi.check(lib.etainv_op_gemm_gnstat(capi.ptr(a), capi.ptr(w), capi.ptr(bias), capi.ptr(res), capi.ptr(x), capi.ptr(part), C.byref(wm), m, c, k, hw,
                                         capi.dtype_code(dtype), capi.stream_ptr()))
    ref = a.float() @ w.float().t() + bias + (res.float() if residual else 0)
    assert relerr(x, ref) < TOL[dtype]
    if wm.value > 0:       # partials against the sums of the STORED values
        pv = part[: m // wm.value * 2 * c].view(m // wm.value, 2, c)
        xs = x.float().view(m // wm.value, wm.value, c)
        torch.testing.assert_close(pv[:, 0], xs.sum(1), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(pv[:, 1], (xs * xs).sum(1), rtol=1e-4, atol=1e-3)
    return x, part, wm.value


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("b,hw,c1,c2,silu,expect", [
    (8, 4096, 320, 0, 1, (64, 0)),          # 256 x 160 ring producer, 64-row blocks
    (4, 1024, 640, 320, 1, None),           # the decoder's concat: two producers (with different tile shapes)
    (16, 1024, 640, 0, 0, (64, 0)),
    (2, 256, 1280, 0, 1, None),             # small tiles
    (1, 64, 1280, 1280, 1, None),
])
def test_groupnorm_from_producer_partials(capi, dtype, b, hw, c1, c2, silu, expect):
    lib = capi.load()
    x1, p1, w1 = _gemm_gnstat(capi, dtype, b, hw, c1, 320, 10, True)
    x2, p2, w2 = _gemm_gnstat(capi, dtype, b, hw, c2, 640, 20, False) if c2 else (None, None, 0)
    if expect is not None:
        assert (w1, w2) == expect
    assert w1 > 0 and (not c2 or w2 > 0)
    C_ = c1 + c2
    gamma, beta = rnd(C_, seed=2) * 0.1 + 1, rnd(C_, seed=3) * 0.1
    out = torch.empty(b, hw, C_, dtype=dtype, device="cuda")
    final = torch.empty(b * (32 * 2 + 2 * C_), dtype=torch.float32, device="cuda")   # (mean, rstd) per group, then the scale / shift planes
    eps = 1e-5 if silu else 1e-6
    capi.check(lib.etainv_op_groupnorm_pre(capi.ptr(x1), capi.ptr(x2), c1, c2, capi.ptr(p1), w1, capi.ptr(p2), w2, capi.ptr(gamma), capi.ptr(beta),
                                           capi.ptr(out), b, hw, 32, eps, silu, capi.ptr(final), capi.dtype_code(dtype), capi.stream_ptr()))
    x = torch.cat([x1.view(b, hw, c1)] + ([x2.view(b, hw, c2)] if c2 else []), dim=-1).float()
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    assert relerr(out, ref.permute(0, 2, 1)) < TOL[dtype]
    g = x.view(b, hw, 32, C_ // 32).permute(0, 2, 1, 3).reshape(b, 32, -1)
    torch.testing.assert_close(final[: b * 64].view(b, 32, 2)[..., 0], g.mean(-1), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(final[: b * 64].view(b, 32, 2)[..., 1], (g.var(-1, unbiased=False) + eps).rsqrt(), rtol=1e-4, atol=1e-6)


def test_gemm_gnstat_ragged_image_reports_none(capi):
    """an image whose pixel count is not a multiple of the wave tile: no partials (the engine runs the statistics pass)"""
    _, _, wm = _gemm_gnstat(capi, torch.float16, 3, 36, 320, 320, 30, True)
    assert wm == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,c", [(4096, 320), (1023, 640), (130, 1280)])
def test_layernorm(capi, dtype, rows, c):
    lib = capi.load()
    x = (rnd(rows, c, seed=1) * 2 + 0.5).to(dtype)
    gamma, beta = rnd(c, seed=2) * 0.1 + 1, rnd(c, seed=3) * 0.1
    out = torch.empty_like(x)
    capi.check(lib.etainv_op_layernorm(capi.ptr(x), capi.ptr(gamma), capi.ptr(beta), capi.ptr(out), rows, c, 1e-5, capi.dtype_code(dtype),
                                       capi.stream_ptr()))
    assert relerr(out, F.layer_norm(x.float(), (c,), gamma, beta, 1e-5)) < TOL[dtype]


# ----------------------------------------------------------------------------------------- attention
def ref_self_attention(qkv, heads, qmap=None, kmap=None, vmap=None):
    b, n, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    q, k, v = qkv.float().split(c, dim=-1)
    idx = torch.arange(b)
    q = q[idx if qmap is None else qmap]
    k = k[idx if kmap is None else kmap]
    v = v[idx if vmap is None else vmap]
    sp = lambda t: t.reshape(b, n, heads, d).permute(0, 2, 1, 3)
    a = (sp(q) @ sp(k).transpose(-1, -2) * d ** -0.5).softmax(-1)
    return (a @ sp(v)).permute(0, 2, 1, 3).reshape(b, n, c)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,d", [(4096, 40), (1024, 80), (256, 160), (64, 160), (144, 160), (576, 80)])
def test_self_attention_plain(capi, dtype, n, d):
    lib = capi.load()
    b, heads = 2, 8
    qkv = rnd(b, n, 3 * heads * d, seed=1, dtype=dtype)
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, 0, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    assert relerr(out, ref_self_attention(qkv, heads)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,b,heads,gain", [(256, 2, 8, 1.0), (64, 1, 8, 1.0), (144, 1, 4, 1.0), (200, 3, 4, 1.0), (576, 1, 8, 1.0), (1024, 2, 8, 3.0),
                                            (2304, 1, 8, 0.05), (9216, 1, 8, 1.0)])
def test_self_attention_d40(capi, dtype, n, b, heads, gain):
    """head_dim 40 kernel (32x32x16 MFMAs, deferred reference maximum): token counts of every latent size incl. 96^2 = 9216, ragged
    key tiles (144, 200), grids with and without the XCD remap (b * heads % 8), and score scales that make the rare
    move-the-maximum branch fire in most tiles (gain 3: logits with a standard deviation of ~9 nats) or never after the first (0.05)."""
    lib = capi.load()
    d = 40
    qkv = rnd(b, n, 3 * heads * d, seed=n + b, dtype=dtype)
    qkv[..., : 2 * heads * d] *= gain
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, 0, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    assert torch.isfinite(out).all()
    assert relerr(out, ref_self_attention(qkv, heads)) < TOL[dtype] * (2 if gain > 1 else 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,b,heads,gain", [(256, 2, 8, 1.0), (144, 1, 4, 1.0), (200, 3, 4, 1.0), (1024, 2, 8, 3.0), (2304, 1, 8, 0.05)])
def test_self_attention_d80(capi, dtype, n, b, heads, gain):
    """head_dim 80 on the same 32x32x16 kernel (one 32-query block per wave, three 32-row tiles of O^T, no zero image)"""
    lib = capi.load()
    d = 80
    qkv = rnd(b, n, 3 * heads * d, seed=n + b + 1, dtype=dtype)
    qkv[..., : 2 * heads * d] *= gain
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, 0, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    assert torch.isfinite(out).all()
    assert relerr(out, ref_self_attention(qkv, heads)) < TOL[dtype] * (2 if gain > 1 else 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,b,heads,gain,mode", [(256, 4, 8, 1.0, 0), (64, 2, 8, 1.0, 0), (144, 1, 4, 1.0, 0), (200, 3, 4, 3.0, 0), (1024, 1, 8, 0.05, 0),
                                                 (256, 8, 8, 1.0, 1), (256, 8, 8, 1.0, 2)])
def test_self_attention_d160(capi, dtype, n, b, heads, gain, mode):
    """head_dim 160 (the (L/4)^2 level) on the same 32x32x16 kernel: six 32-row tiles of O^T, eleven K slices per score tile, one block per CU; ragged last
    key tiles, the remap modes (n_img = 2), large and small score scales.  ETAINV_ATT160_V2=0 is the generic 16x16x32 kernel it replaces there."""
    lib = capi.load()
    d, n_img = 160, 2 if mode else 1
    qkv = rnd(b, n, 3 * heads * d, seed=n + b + 2, dtype=dtype)
    qkv[..., : 2 * heads * d] *= gain
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, mode, n_img, capi.dtype_code(dtype), capi.stream_ptr()))
    ident = torch.arange(b)
    qm, km, vm = ident.clone(), ident.clone(), ident.clone()
    for img in range(n_img if mode else 0):
        u_s, u_t, c_s, c_t = img, n_img + img, 2 * n_img + img, 3 * n_img + img
        if mode == 1:
            qm[c_t], km[c_t] = c_s, c_s
        else:
            km[u_t], vm[u_t], km[c_t], vm[c_t] = u_s, u_s, c_s, c_s
    assert torch.isfinite(out).all()
    assert relerr(out, ref_self_attention(qkv, heads, qm, km, vm)) < TOL[dtype] * (2 if gain > 1 else 1)


def test_self_attention_d40_maximum_jumps_late(capi):
    """A key whose score exceeds everything before it by far more than the deferral threshold, placed in a LATE tile, for a few queries
    only (the branch is wave-uniform, the update per query), plus a first tile whose scores are all very negative for other queries."""
    lib = capi.load()
    b, heads, n, d, dtype = 1, 8, 512, 40, torch.float16
    qkv = rnd(b, n, 3 * heads * d, seed=77, dtype=dtype)
    q, k = qkv[..., : heads * d], qkv[..., heads * d: 2 * heads * d]
    k[0, 300] = (q[0, 5] * 6).to(dtype)          # key 300 (tile 4) aligned with query 5: a score far above its running maximum
    k[0, 450] = (q[0, 130] * 8).to(dtype)
    k[0, :64] = (k[0, :64] - 4 * q[0, 200:201]).to(dtype)   # first tile strongly negative for query 200
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, 0, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    ref = ref_self_attention(qkv, heads)
    assert torch.isfinite(out).all()
    assert relerr(out, ref) < TOL[dtype]
    assert relerr(out[0, [5, 130, 200]], ref[0, [5, 130, 200]]) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_self_attention_d40_single_row_call_equals_the_row_of_a_batch(capi, dtype, monkeypatch):
    """Round 6: a launch of <= 256 blocks (single-image calls) runs one 32-query block per wave instead of two -- the same arithmetic per query block: a row
    computed alone must be bit-equal to the same row inside a 16-row call."""
    lib = capi.load()
    b, heads, n, d = 16, 8, 4096, 40
    qkv = rnd(b, n, 3 * heads * d, seed=9, dtype=dtype)
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    monkeypatch.setenv("ETAINV_A40_PERSIST", "0")   # (16 rows would otherwise take the persistent kernel, whose first tile is rounded differently)
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, 0, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    one_in, one = qkv[3:4].contiguous(), torch.empty(1, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(one_in), capi.ptr(one), 1, n, heads, d, 0, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    assert torch.equal(one[0], out[3])
    assert relerr(one, ref_self_attention(one_in, heads)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_self_attention_d40_speculative_maximum(capi, dtype):
    """Round 6: from 8 key tiles on, only tile 0 computes the reference maximum; later tiles exponentiate against it, and a block whose denominators come out
    non-finite repeats its pass with the running maximum (attention.hip, self_attn40_kernel).  N = 2048 (32 tiles), four regimes in ONE launch, each in its own
    (batch row, head) so that blocks of both kinds run side by side:  row 0 plain;  row 1: late keys 2^6 .. 2^12 above the first tile's maximum for some queries
    (the tracked pass would move m', the speculative one keeps P large: no overflow in either dtype);  row 2: one late key ~2^40 above it (fp16 P overflows ->
    fallback; bf16 stays finite);  row 3: the first tile far BELOW everything else for every query (all of P large)."""
    lib = capi.load()
    b, heads, n, d = 4, 8, 2048, 40
    qkv = rnd(b, n, 3 * heads * d, seed=123, dtype=dtype)
    q, k = qkv[..., : heads * d], qkv[..., heads * d: 2 * heads * d]
    sc = 1.0 / (d ** 0.5)
    def boost(row, head, key, query, log2_gain):
        # make score(query, key) of (row, head) ~ log2_gain * ln 2 above that query's typical maximum: k := q * t with t = gain / (|q|^2 * scale)
        qv = q[row, query, head * d:(head + 1) * d].float()
        t = (log2_gain * 0.6931 + 6.0) / (float(qv @ qv) * sc)
        k[row, key, head * d:(head + 1) * d] = (qv * t).to(dtype)
    for i, (key, g) in enumerate([(700, 6), (1300, 9), (2000, 12), (1999, 8)]):
        boost(1, i % heads, key, 64 * i + 7, g)
    boost(2, 3, 1500, 300, 40)
    k[3, :64] = (k[3, :64] * 0.02).to(dtype)
    q[3] = (q[3] * 3).to(dtype)                      # sharper rows: the first tile's maximum sits well below the later tiles'
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, 0, 1, capi.dtype_code(dtype), capi.stream_ptr()))
    ref = ref_self_attention(qkv, heads)
    assert torch.isfinite(out).all()
    for row in range(b):
        assert relerr(out[row], ref[row]) < 1.5 * TOL[dtype], row
    assert relerr(out[2, 300, 3 * d:4 * d], ref[2, 300, 3 * d:4 * d]) < 2 * TOL[dtype]     # the query whose key forced the fallback pass in fp16


def _self40(capi, qkv, b, n, heads, mode=0, n_img=1, d=40):
    lib = capi.load()
    out = torch.empty(b, n, heads * d, dtype=qkv.dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, mode, n_img, capi.dtype_code(qkv.dtype), capi.stream_ptr()))
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,b,heads", [(2048, 16, 8), (1024, 33, 8), (2048, 33, 4), (4096, 9, 8)])
def test_self_attention_d40_persistent_kernel(capi, dtype, n, b, heads, monkeypatch):
    """Round 6: launches with at least two (row, head, 512-query block) items per CU run self_attn40q_kernel (one block per CU walks the items: attention.hip).  Item
    counts that divide the 256 blocks evenly (512) and that do not (528, 576: some blocks run one item more; the K / V stream of a block's last item has no successor),
    head counts with and without the XCD-aware item order (b * heads % 8).  Against the fp32 reference and against the kernel it replaces."""
    qkv = rnd(b, n, 3 * heads * 40, seed=n + b, dtype=dtype)
    out = _self40(capi, qkv, b, n, heads)
    monkeypatch.setenv("ETAINV_A40_PERSIST", "0")
    old = _self40(capi, qkv, b, n, heads)
    ref = ref_self_attention(qkv, heads)
    assert torch.isfinite(out).all()
    assert relerr(out, ref) < TOL[dtype]
    assert abs(relerr(out, ref) - relerr(old, ref)) < 0.1 * TOL[dtype]       # the same precision as the kernel it replaces ...
    assert not torch.equal(out, old) or n < 1024                               # ... which is another kernel (the switch works)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,b,heads,mode", [(1024, 16, 8, 0), (2304, 8, 8, 0), (1024, 33, 4, 0), (1024, 16, 8, 1), (1024, 16, 8, 2)])
def test_self_attention_d80_persistent_kernel(capi, dtype, n, b, heads, mode, monkeypatch):
    """The same kernel at head_dim 80 (two query blocks per wave, items of 256 queries; N = 2304 is the 768^2 image's level): even and uneven item counts, both item orders,
    the prompt-to-prompt / MasaCtrl row couplings (n_img = 4).  Against the fp32 reference and the kernel it replaces (ETAINV_A80_PERSIST=0)."""
    d, n_img = 80, 4 if mode else 1
    qkv = rnd(b, n, 3 * heads * d, seed=n + b + mode, dtype=dtype)
    out = _self40(capi, qkv, b, n, heads, mode, n_img, d)
    monkeypatch.setenv("ETAINV_A80_PERSIST", "0")
    old = _self40(capi, qkv, b, n, heads, mode, n_img, d)
    ident = torch.arange(b)
    qm, km, vm = ident.clone(), ident.clone(), ident.clone()
    for img in range(n_img if mode else 0):
        u_s, u_t, c_s, c_t = img, n_img + img, 2 * n_img + img, 3 * n_img + img
        if mode == 1:
            qm[c_t], km[c_t] = c_s, c_s
        else:
            km[u_t], vm[u_t], km[c_t], vm[c_t] = u_s, u_s, c_s, c_s
    ref = ref_self_attention(qkv, heads, qm, km, vm)
    assert torch.isfinite(out).all()
    assert relerr(out, ref) < TOL[dtype]
    assert abs(relerr(out, ref) - relerr(old, ref)) < 0.1 * TOL[dtype]
    assert not torch.equal(out, old)


@pytest.mark.parametrize("dtype", DTYPES)
def test_self_attention_d40_persistent_kernel_exact_pass(capi, dtype):
    """The four regimes of test_self_attention_d40_speculative_maximum inside a 16-row launch of the persistent kernel: an item whose denominators overflow (fp16,
    row 2) is repeated with the running maximum INSIDE the item loop, and the block then restarts its K / V stream for the next item."""
    b, heads, n, d = 16, 8, 2048, 40
    qkv = rnd(b, n, 3 * heads * d, seed=321, dtype=dtype)
    q, k = qkv[..., : heads * d], qkv[..., heads * d: 2 * heads * d]
    sc = 1.0 / (d ** 0.5)
    def boost(row, head, key, query, log2_gain):
        qv = q[row, query, head * d:(head + 1) * d].float()
        t = (log2_gain * 0.6931 + 6.0) / (float(qv @ qv) * sc)
        k[row, key, head * d:(head + 1) * d] = (qv * t).to(dtype)
    for i, (key, g) in enumerate([(700, 6), (1300, 9), (2000, 12), (1999, 8)]):
        boost(1, i % heads, key, 64 * i + 7, g)
    for row, head, key, query in [(2, 3, 1500, 300), (7, 0, 100, 1999), (15, 7, 2047, 0)]:   # fp16: three items (first, middle, last query block) take the exact pass
        boost(row, head, key, query, 40)
    k[3, :64] = (k[3, :64] * 0.02).to(dtype)
    q[3] = (q[3] * 3).to(dtype)
    out = _self40(capi, qkv, b, n, heads)
    ref = ref_self_attention(qkv, heads)
    assert torch.isfinite(out).all()
    for row in range(b):
        assert relerr(out[row], ref[row]) < 1.5 * TOL[dtype], row
    for row, head, query in [(2, 3, 300), (7, 0, 1999), (15, 7, 0)]:
        assert relerr(out[row, query, head * d:(head + 1) * d], ref[row, query, head * d:(head + 1) * d]) < 2 * TOL[dtype], (row, head, query)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_self_attention_d40_persistent_kernel_remaps(capi, mode, dtype):
    """prompt-to-prompt self-replace (mode 1) and MasaCtrl (mode 2) row couplings through the persistent kernel's item decoder (n_img = 4: 16 rows, 512 items)"""
    n_img, heads, n = 4, 8, 2048
    b = 4 * n_img
    qkv = rnd(b, n, 3 * heads * 40, seed=5 + mode, dtype=dtype)
    out = _self40(capi, qkv, b, n, heads, mode, n_img)
    ident = torch.arange(b)
    qm, km, vm = ident.clone(), ident.clone(), ident.clone()
    for img in range(n_img):
        u_s, u_t, c_s, c_t = img, n_img + img, 2 * n_img + img, 3 * n_img + img
        if mode == 1:
            qm[c_t], km[c_t] = c_s, c_s
        else:
            km[u_t], vm[u_t], km[c_t], vm[c_t] = u_s, u_s, c_s, c_s
    assert relerr(out, ref_self_attention(qkv, heads, qm, km, vm)) < TOL[dtype]


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_self_attention_d40_remaps(capi, mode, dtype):
    lib = capi.load()
    n_img, heads, n, d = 2, 8, 320, 40
    b = 4 * n_img
    qkv = rnd(b, n, 3 * heads * d, seed=3, dtype=dtype)
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, mode, n_img, capi.dtype_code(dtype), capi.stream_ptr()))
    ident = torch.arange(b)
    qm, km, vm = ident.clone(), ident.clone(), ident.clone()
    for img in range(n_img):
        u_s, u_t, c_s, c_t = img, n_img + img, 2 * n_img + img, 3 * n_img + img
        if mode == 1:
            qm[c_t] = c_s
            km[c_t] = c_s
        else:
            km[u_t], vm[u_t] = u_s, u_s
            km[c_t], vm[c_t] = c_s, c_s
    assert relerr(out, ref_self_attention(qkv, heads, qm, km, vm)) < TOL[dtype]


@pytest.mark.parametrize("mode", [1, 2])
def test_self_attention_remaps(capi, mode):
    """mode 1 = prompt-to-prompt self-replace (ptp.py:194-199), mode 2 = MasaCtrl (masactrl.py:56-72), n_img = 2."""
    lib = capi.load()
    n_img, heads, n, d, dtype = 2, 8, 256, 80, torch.float16
    b = 4 * n_img
    qkv = rnd(b, n, 3 * heads * d, seed=2, dtype=dtype)
    out = torch.empty(b, n, heads * d, dtype=dtype, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), b, n, heads, d, mode, n_img, capi.F16, capi.stream_ptr()))
    ident = torch.arange(b)
    qm, km, vm = ident.clone(), ident.clone(), ident.clone()
    for img in range(n_img):
        u_s, u_t, c_s, c_t = img, n_img + img, 2 * n_img + img, 3 * n_img + img
        if mode == 1:
            qm[c_t] = c_s
            km[c_t] = c_s
        else:
            km[u_t], vm[u_t] = u_s, u_s
            km[c_t], vm[c_t] = c_s, c_s
    assert relerr(out, ref_self_attention(qkv, heads, qm, km, vm)) < TOL[dtype]


def test_masactrl_vs_reference_golden(capi, golden):
    """The reference's MutualSelfAttentionControl outputs (tests/golden/masactrl.npz, head_dim 8) cannot be fed to the
    d in {40,80,160} kernels directly; the oracle's MasaCtrl (pinned by that fixture) is used at d = 40 instead."""
    from oracle import loop as oloop
    lib = capi.load()
    heads, n, d = 8, 64, 40
    qkv = rnd(4, n, 3 * heads * d, seed=9, dtype=torch.float16)
    out = torch.empty(4, n, heads * d, dtype=torch.float16, device="cuda")
    capi.check(lib.etainv_op_self_attention(capi.ptr(qkv), capi.ptr(out), 4, n, heads, d, 2, 1, capi.F16, capi.stream_ptr()))
    q, k, v = qkv.float().cpu().split(heads * d, dim=-1)
    hb = lambda t: t.reshape(4, n, heads, d).permute(0, 2, 1, 3).reshape(4 * heads, n, d)
    m = oloop.MasaCtrl(4, 10)
    m.cur_step, m.cur_att_layer = 4, 20
    ref = m(False, 20, "up", hb(q), hb(k), hb(v), d ** -0.5, heads)
    assert relerr(out.cpu(), ref) < TOL[torch.float16]


def _ptp_tables(n_img):
    from oracle import ptp as optp
    pairs = json.load(open(__file__.rsplit("/", 1)[0] + "/golden/prompt_pairs.json"))
    tok = optp.WordTokenizer()
    mp, al, eq, ca = [], [], [], []
    for i in range(n_img):
        src, tgt = pairs[i % 4]
        m, a = optp.refinement_mapper(src, tgt, tok)
        mp.append(m)
        al.append(a)
        eq.append(optp.equalizer(tgt, (tgt.split(" ")[1],), (2.0,), tok))
        ca.append(optp.time_words_alpha([src, tgt], 10, {"default_": 0.4}, tok)[0, 0])
    T = lambda x, dt: torch.tensor(np.stack(x), dtype=dt).cuda()
    return T(mp, torch.int32), T(al, torch.float32), T(eq, torch.float32), T(ca, torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,d", [(256, 160), (1024, 80), (4096, 40), (64, 160)])
def test_cross_attention_ptp_edit_and_store(capi, dtype, n, d):
    """Cross-attention with the fused Refine + Reweight edit on the cond target rows and AttentionStore accumulation,
    against the oracle's controller algebra (pinned by tests/golden/ptp_algebra.npz) on materialised probabilities."""
    from oracle import ptp as optp
    lib = capi.load()
    n_img, heads = 2, 8
    b = 4 * n_img
    c = heads * d
    q = rnd(b, n, c, seed=1, dtype=dtype)
    kv = rnd(b, 77, 2 * c, seed=2, dtype=dtype)
    mapper, alphas, eq, ca = _ptp_tables(n_img)
    out = torch.empty(b, n, c, dtype=dtype, device="cuda")
    store = n == 256
    maps = torch.zeros(5, n_img, 2, heads, n, 77, dtype=torch.float32, device="cuda") if store else None
    ctrl = capi.AttnCtrl(mode=capi.ATTN_PTP, n_img=n_img, store_maps=int(store), mapper=capi.ptr(mapper), alphas=capi.ptr(alphas),
                         equalizer=capi.ptr(eq), cross_alpha=capi.ptr(ca))
    for _ in range(2):  # two "steps" so accumulation is exercised
        capi.check(lib.etainv_op_cross_attention(capi.ptr(q), capi.ptr(kv), capi.ptr(out), b, n, heads, d, 77, C.byref(ctrl), 3, n_img,
                                                 capi.ptr(maps), capi.dtype_code(dtype), capi.stream_ptr()))
    # reference: probabilities -> oracle controller (one per image, rows [u_s,u_t,c_s,c_t]) -> P @ V
    k, v = kv.float().split(c, dim=-1)
    sp = lambda t, L_: t.float().reshape(b, L_, heads, d).permute(0, 2, 1, 3)
    probs = (sp(q, n) @ sp(k, 77).transpose(-1, -2) * d ** -0.5).softmax(-1).cpu()       # (b, heads, n, 77)
    ref_maps = torch.zeros(n_img, 2, heads, n, 77)
    for img in range(n_img):
        rows = [img, n_img + img, 2 * n_img + img, 3 * n_img + img]
        ctl = optp.AttentionEdit(10, ca[img].cpu().numpy()[None, None].repeat(11, 0), 0.6, mapper=mapper[img].cpu().numpy().astype(np.int64),
                                 alphas=alphas[img].cpu().numpy(), equalizer=eq[img].cpu().numpy(), num_att_layers=1)
        attn = probs[rows].reshape(4 * heads, n, 77).clone()
        attn = ctl(attn, True, "up")
        probs[rows] = attn.reshape(4, heads, n, 77)
        ref_maps[img, 0] = probs[rows[2]]
        ref_maps[img, 1] = probs[rows[3]]
    ref = (probs.cuda() @ sp(v, 77)).permute(0, 2, 1, 3).reshape(b, n, c)
    assert relerr(out, ref) < TOL[dtype]
    if store:
        assert relerr(maps[3].cpu(), 2 * ref_maps) < 1e-3
        assert float(maps[0].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_cross_attention_xcd_head_placement_is_bit_identical(capi, dtype, monkeypatch):
    """Round 6: the launch deals its block ids so that the eight heads of a (row, query range) run on one XCD (CrossParams::xcd_gx) -- placement only: the prompt-to-prompt call of
    the test above (plain launch over three row groups + edit launch over the cond-target rows, with the map store) must give the same bits as the round-5 placement."""
    lib = capi.load()
    n_img, heads, n, d = 4, 8, 1024, 80
    b, c = 4 * n_img, heads * d
    q = rnd(b, n, c, seed=11, dtype=dtype)
    kv = rnd(b, 77, 2 * c, seed=12, dtype=dtype)
    mapper, alphas, eq, ca = _ptp_tables(n_img)
    outs, stores = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("ETAINV_CROSS_XCD", flag)
        out = torch.empty(b, n, c, dtype=dtype, device="cuda")
        maps = torch.zeros(5, n_img, 2, heads, n, 77, dtype=torch.float32, device="cuda")
        ctrl = capi.AttnCtrl(mode=capi.ATTN_PTP, n_img=n_img, store_maps=1, mapper=capi.ptr(mapper), alphas=capi.ptr(alphas), equalizer=capi.ptr(eq), cross_alpha=capi.ptr(ca))
        capi.check(lib.etainv_op_cross_attention(capi.ptr(q), capi.ptr(kv), capi.ptr(out), b, n, heads, d, 77, C.byref(ctrl), 2, n_img, capi.ptr(maps), capi.dtype_code(dtype),
                                                 capi.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
        stores.append(maps)
    assert torch.equal(outs[0], outs[1]) and torch.equal(stores[0], stores[1])
    assert float(stores[0][2].abs().max()) > 0.0


def test_cross_attention_store_layout_forward(capi):
    """forward-pass layout [u x B, c x B]: only cond rows are stored, role 0."""
    lib = capi.load()
    n_img, heads, n, d = 2, 8, 256, 160
    for rows in (2 * n_img, n_img):
        q = rnd(rows, n, heads * d, seed=1, dtype=torch.float16)
        kv = rnd(rows, 77, 2 * heads * d, seed=2, dtype=torch.float16)
        out = torch.empty_like(q)
        maps = torch.zeros(5, n_img, 2, heads, n, 77, dtype=torch.float32, device="cuda")
        ctrl = capi.AttnCtrl(mode=capi.ATTN_STORE, n_img=n_img, store_maps=1)
        capi.check(lib.etainv_op_cross_attention(capi.ptr(q), capi.ptr(kv), capi.ptr(out), rows, n, heads, d, 77, C.byref(ctrl), 0, n_img,
                                                 capi.ptr(maps), capi.F16, capi.stream_ptr()))
        k = kv.float()[..., :heads * d]
        sp = lambda t, L_: t.float().reshape(rows, L_, heads, d).permute(0, 2, 1, 3)
        probs = (sp(q, n) @ sp(k, 77).transpose(-1, -2) * d ** -0.5).softmax(-1)
        cond = probs[rows - n_img:]
        assert relerr(maps[0, :, 0], cond) < 1e-3
        assert float(maps[0, :, 1].abs().max()) == 0.0


# ----------------------------------------------------------------------------------------- map consumers
def test_word_maps_and_local_blend_vs_oracle(capi):
    from oracle import ptp as optp
    lib = capi.load()
    n_img, heads, res, L = 2, 8, 16, 64
    g = torch.Generator().manual_seed(5)
    acc = torch.rand(5, n_img, 2, heads, res * res, 77, generator=g) ** 4 * 3.0       # "sums over 3 steps"
    acc_d = acc.cuda()
    tokens = torch.tensor([[1, 2, 5], [3, 1, 4]], dtype=torch.int32).cuda()
    out = torch.zeros(n_img, 3, L, L, device="cuda")
    capi.check(lib.etainv_op_word_maps(capi.ptr(acc_d), 5, n_img, heads, res, L, n_img, capi.ptr(tokens), 3, 3, capi.ptr(out), 0, 1.0,
                                       capi.stream_ptr()))
    for img in range(n_img):
        st = optp.AttentionStore()
        st.cur_step = 3
        lay = [acc[l, img, 0].clone() for l in range(5)]                              # role 0 = forward-pass cond rows
        st.attention_store = {"down_cross": lay[:2], "up_cross": lay[2:], "mid_cross": [], "down_self": [], "mid_self": [], "up_self": []}
        for j in range(3):
            ref = optp.attention_map(st, int(tokens[img, j]), res=16, resize=64)
            torch.testing.assert_close(out[img, j].cpu(), ref[0], rtol=1e-4, atol=1e-5)
    # LocalBlend
    x = torch.randn(2 * n_img, 4, L, L, generator=g)
    alpha = torch.zeros(n_img, 2, 77)
    alpha[0, 0, 2] = alpha[0, 1, 2] = 1
    alpha[1, 0, 3] = alpha[1, 1, 4] = 1
    xd = x.clone().cuda()
    ad = alpha.cuda()
    capi.check(lib.etainv_op_local_blend(capi.ptr(acc_d), 5, n_img, heads, res, L, capi.ptr(xd), n_img, capi.ptr(ad), 0.3, capi.stream_ptr()))
    for img in range(n_img):
        lb = optp.LocalBlend(alpha[img].numpy(), 10, res=16)
        lb.counter = 100
        lay = [acc[l, img].reshape(2 * heads, res * res, 77) for l in range(5)]
        store = {"down_cross": [None, None, lay[0], lay[1]], "up_cross": lay[2:]}
        ref = lb(torch.stack([x[img], x[n_img + img]]), store)
        torch.testing.assert_close(xd[img].cpu(), ref[0], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(xd[n_img + img].cpu(), ref[1], rtol=1e-6, atol=1e-6)
    # round 6: the partial dot products of a (role, pixel) item are split over several threads and summed in the old order -- the same bits as the unsplit loop
    import os
    x2 = x.clone().cuda()
    os.environ["ETAINV_BLEND_NOSPLIT"] = "1"
    try:
        capi.check(lib.etainv_op_local_blend(capi.ptr(acc_d), 5, n_img, heads, res, L, capi.ptr(x2), n_img, capi.ptr(ad), 0.3, capi.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        del os.environ["ETAINV_BLEND_NOSPLIT"]
    assert torch.equal(x2, xd)
