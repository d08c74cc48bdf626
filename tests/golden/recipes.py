"""Seeded input recipes shared by make_golden.py (which feeds them to the reference) and the tests
(which feed them to the oracle / the HIP path).  Only OUTPUTS are stored in the npz fixtures;
inputs are regenerated from these recipes and guarded by the crc32 values stored beside them."""
import zlib
import numpy as np
import torch

ETA_CASES = {  # name: (eta spec, timestep, fp16?, use_mask)
    "lin_t980": ((0.0, 0.4), 980, False, True),
    "lin_t0": ((0.0, 0.4), 0, False, True),
    "paper_t600": ([[0.6, 0], [1, 0.7]], 600, False, True),       # eta == 0 exactly -> 0/0 (SURVEY E-7)
    "paper_t620": ([[0.6, 0], [1, 0.7]], 620, False, True),       # tiny eta
    "paper_t980": ([[0.6, 0], [1, 0.7]], 980, False, True),
    "paper_t620_f16": ([[0.6, 0], [1, 0.7]], 620, True, True),    # fp16 overflow hazard
    "paper_t980_nomask": ([[0.6, 0], [1, 0.7]], 980, False, False),
}


ETA_MODE_CASES = {  # name: mask_mode_cfg overrides (eta_inversion.py:88-101,164-201); all at paper eta, t = 980, fp32
    "gt_thres": dict(mask_eta="gt", thres=0.5),
    "fwd_t": dict(mask_eta="fwd", thres=0.2),
    "soft": dict(mask_eta="fwd_mean", thres=None),
    "soft_pow": dict(mask_eta="fwd_mean", thres=None, pow=2.0),
    "thres_pow": dict(mask_eta="fwd_mean", thres=0.3, pow=3.0),
}


ETA_DIRINV_CASES = {  # target_dirinv (eta_inversion.py:251-256) with / without a mask_dirinv; paper eta, t = 980, fp32
    "tdir": dict(mask_eta="fwd_mean", thres=0.2, target_dirinv=0.5),
    "tdir_masked": dict(mask_eta="fwd_mean", mask_dirinv="fwd_mean", thres=0.4, target_dirinv=0.8),
    "tdir_soft": dict(mask_eta="fwd_mean", mask_dirinv="fwd_mean", thres=None, pow=2.0, target_dirinv=0.3),
    # mask_dirinv from a DIFFERENT source than mask_eta (eta_inversion.py:234-236): the `mask` argument (gt) is the eta map rolled by 7 pixels
    "tdir_gt": dict(mask_eta="fwd_mean", mask_dirinv="gt", thres=0.4, target_dirinv=0.6),
    "tdir_gt_eta_fwd": dict(mask_eta="gt", mask_dirinv="fwd", thres=0.3, pow=2.0, target_dirinv=0.7),
}


def dirinv_gt_mask(mask_map: torch.Tensor) -> torch.Tensor:
    """the ground-truth `mask` argument of the ETA_DIRINV_CASES (differs from the forward maps so that a swapped source shows)"""
    return torch.roll(mask_map, 7, dims=-1)


def eta_case_inputs(name: str, L: int = 64):
    if name in ETA_MODE_CASES or name in ETA_DIRINV_CASES:
        return _eta_case_inputs("mode_" + name, False, L)
    return _eta_case_inputs(name, ETA_CASES[name][2], L)


def _eta_case_inputs(name: str, fp16: bool, L: int = 64):
    dt = torch.float16 if fp16 else torch.float32
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0xFFFF)
    latent = torch.randn(2, 4, L, L, generator=g).to(dt)
    unet_out = torch.randn(4, 4, L, L, generator=g).to(dt)       # rows [u_s, u_t, c_s, c_t]
    src_prev = (latent[:1].float() * 0.98 + 0.05 * torch.randn(1, 4, L, L, generator=g)).to(dt)
    mask_map = torch.rand(1, L, L, generator=g).to(dt)
    noise = torch.randn((10, 1, 4, L, L), generator=torch.Generator().manual_seed(5)).to(dt)
    return dict(latent=latent, unet_out=unet_out, src_prev=src_prev, mask_map=mask_map, noise=noise)


def crc(t: torch.Tensor) -> int:
    return zlib.crc32(t.detach().float().contiguous().numpy().tobytes())


# ------------------------------------------------------------------ prompt-to-prompt controller drive
def attention_order():
    """The 32 attention calls of one SD1.x UNet forward at 64x64 latents: (place, is_cross, N)."""
    order = []
    for place, ns in (("down", [4096, 4096, 1024, 1024, 256, 256]), ("mid", [64]),
                      ("up", [256, 256, 256, 1024, 1024, 1024, 4096, 4096, 4096])):
        for n in ns:
            order += [(place, False, n), (place, True, n)]
    assert len(order) == 32
    return order


def _rand_probs(gen, rows, n, m):
    return torch.softmax(2.0 * torch.randn(rows, n, m, generator=gen), dim=-1)


def _layer_input(gen, rows, is_cross, n):
    # 64^2 self maps are never edited nor stored (reference ptp.py:153,195): a small stand-in with N > 32^2
    if is_cross or n <= 1024:
        return _rand_probs(gen, rows, n, 77 if is_cross else n)
    return _rand_probs(gen, rows, 1025, 8)


def summarize(t: torch.Tensor) -> torch.Tensor:
    """Weighted sum over the key axis (weights 0.5..1.5): compact, order-sensitive digest of a map."""
    w = torch.linspace(0.5, 1.5, t.shape[-1], dtype=torch.float64)
    return (t.double() * w).sum(-1).float()


def drive_edit_controller(ctrl, steps=4, heads=2, seed=123):
    """Feed seeded random probabilities through `ctrl(attn, is_cross, place)` for `steps` UNet forwards of a
    [u_s,u_t,c_s,c_t] batch and call `ctrl.step_callback(x_t)` after each.  Returns digests of every edited
    layer output (N <= 1024) at the first and last step, one full small layer, and the blended latents."""
    gen = torch.Generator().manual_seed(seed)
    x_t = torch.randn(2, 4, 64, 64, generator=gen)
    out, xs = {}, []
    for step in range(steps):
        for li, (place, is_cross, n) in enumerate(attention_order()):
            attn = _layer_input(gen, 4 * heads, is_cross, n)
            after = ctrl(attn, is_cross, place)
            if step in (0, steps - 1) and n <= 1024:
                out[f"s{step}_l{li}"] = summarize(after)
                if n == 64:
                    out[f"s{step}_l{li}_full"] = after.float().clone()
        x_t = ctrl.step_callback(x_t)
        xs.append(x_t.clone())
        x_t = x_t + 0.1 * torch.randn(2, 4, 64, 64, generator=gen)
    out["x_t"] = torch.stack(xs)
    return out


def drive_store_controller(ctrl_call, get_maps, steps=3, heads=8, seed=321):
    """Forward-pass AttentionStore drive for a [u, c] batch; `get_maps()` returns the per-word maps
    (W,1,64,64) after each step (reference eta_inversion.py:44-49)."""
    gen = torch.Generator().manual_seed(seed)
    maps = []
    for step in range(steps):
        for li, (place, is_cross, n) in enumerate(attention_order()):
            ctrl_call(_layer_input(gen, 2 * heads, is_cross, n), is_cross, place)
        maps.append(get_maps())
    return torch.stack(maps)
