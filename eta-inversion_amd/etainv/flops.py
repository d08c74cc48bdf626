"""Algorithmic MACs of one SD1.x UNet sample-forward by layer walk (401.64 GMAC at L = 64, 1074.06 at L = 96), and the
share of them in front of an early exit of a row after transformer block `k` (etainv_attn_ctrl.src_exit_block).  Host arithmetic only --
used for the loops' row accounting (`EtaLoop.rows_executed`); bench.py reports the FLOPs the kernels actually executed."""
from functools import lru_cache

CH = (320, 640, 1280, 1280)
CTX_LEN, CTX_DIM, TEMB = 77, 768, 1280


def _res(cin, cout, n):
    m = n * 9 * cin * cout + n * 9 * cout * cout + TEMB * cout
    if cin != cout:
        m += n * cin * cout
    return m


def _tb(c, n):
    m = 2 * n * c * c                       # proj_in, proj_out (1x1 convs)
    m += 3 * n * c * c + 2 * n * n * c      # self-attention: fused QKV, QK^T, PV (8 heads x d = c)
    m += n * c * c                          # attn1.to_out
    m += n * c * c + 2 * CTX_LEN * CTX_DIM * c + 2 * n * CTX_LEN * c + n * c * c    # cross: to_q, to_k / to_v of the context, QK^T, PV, to_out
    m += n * c * 8 * c + n * 4 * c * c      # GEGLU projection, FF out
    return m


@lru_cache(maxsize=None)
def unet_macs(L, exit_after_block=None):
    """MACs of one sample-forward at latent side L; with `exit_after_block` = k, only what runs up to and including transformer block k
    (0-based in execution order: 0-5 down, 6 mid, 7-9 up1, 10-12 up2, 13-15 up3)."""
    total, ti = 0, 0
    done = lambda: exit_after_block is not None and ti > exit_after_block
    side = L
    total += side * side * 9 * 4 * CH[0]                               # conv_in
    total += 320 * TEMB + TEMB * TEMB                                  # time MLP
    hc = CH[0]
    skips = [CH[0]]
    for i in range(4):                                                 # down
        for _ in range(2):
            total += _res(hc, CH[i], side * side)
            hc = CH[i]
            if i < 3:
                total += _tb(hc, side * side)
                ti += 1
            skips.append(hc)
        if i < 3:
            side //= 2
            total += side * side * 9 * hc * hc                         # stride-2 conv
            skips.append(hc)
    total += _res(hc, hc, side * side) + _tb(hc, side * side) + _res(hc, hc, side * side)    # mid
    ti += 1
    rev = (1280, 1280, 640, 320)
    for i in range(4):                                                 # up
        for _ in range(3):
            if done():
                return total
            total += _res(hc + skips.pop(), rev[i], side * side)
            hc = rev[i]
            if i > 0:
                total += _tb(hc, side * side)
                ti += 1
        if i < 3:
            if done():
                return total
            side *= 2
            total += side * side * 9 * hc * hc                         # nearest x2 + conv
    if done():
        return total
    return total + L * L * 9 * CH[0] * 4                               # conv_out


def unet_tflop(L):
    return 2 * unet_macs(L) / 1e12


def exit_share(L, block):
    """share of a sample-forward's MACs a row has run when it leaves after transformer block `block`"""
    return unet_macs(L, block) / unet_macs(L)
