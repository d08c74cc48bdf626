"""SD1.x UNet restated for the CPU oracle (test infrastructure, see oracle/__init__.py).

The reference never restates the UNet; it calls `model.unet(latent, t, encoder_hidden_states=ctx)
["sample"]` (modules/inversion/eta_inversion.py:321).  The graph below follows SURVEY.md
Appendix A (diffusers 0.21.1 `UNet2DConditionModel` with the SD1.x config); in-tree evidence:
ResnetBlock2D forward restated at modules/utils/pnp_utils.py:136-185, attention forward at
modules/utils/ptp_utils.py:221-260, module tree at modules/utils/pnp_utils.py:45-58.

Module / parameter names equal the diffusers state-dict keys so that (a) a real SD1.x
safetensors snapshot loads with `load_state_dict`, (b) the reference's hook installers, which
look for descendants whose class name is `Attention` under children named down*/mid*/up*
(ptp_utils.py:277-299, masactrl_utils.py:129-150), find exactly 32 of them in execution order.

`Attention.forward` takes an optional `ctrl` callable -- the oracle's own declarative stand-in
for the reference's monkey-patched forward (ptp_utils.py:205-260): `ctrl(kind, layer_idx, place,
q, k, v, scale, heads)` returns the attention output `(B, N, C)` or None for plain attention.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F


def timestep_embedding(t: torch.Tensor, dim: int = 320, max_period: float = 10000.0) -> torch.Tensor:
    """Sinusoidal embedding, flip_sin_to_cos=True, freq_shift=0."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / half
    freqs = torch.exp(exponent)
    args = t.to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.linear_2 = nn.Linear(cout, cout)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """h = conv1(silu(gn(x))) + temb_proj(silu(temb)); h = conv2(silu(gn(h))); out = shortcut(x) + h
    (pnp_utils.py:139-185)."""

    def __init__(self, cin, cout, temb_ch=1280, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """Class name must be `Attention` (ptp_utils.py:278, masactrl_utils.py:132)."""

    def __init__(self, query_dim, cross_dim=None, heads=8):
        super().__init__()
        self.heads = heads
        self.scale = (query_dim // heads) ** -0.5
        kv_dim = cross_dim if cross_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(kv_dim, query_dim, bias=False)
        self.to_v = nn.Linear(kv_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Identity()])
        # oracle-side bookkeeping (set by UNet): execution index 0..31 and place in unet
        self.layer_idx = -1
        self.place = ""
        self.ctrl = None

    def head_to_batch_dim(self, x):
        b, n, c = x.shape
        h = self.heads
        return x.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def batch_to_head_dim(self, x):
        bh, n, d = x.shape
        h = self.heads
        return x.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, d * h)

    def forward(self, x, encoder_hidden_states=None, attention_mask=None):
        is_cross = encoder_hidden_states is not None
        ctx = encoder_hidden_states if is_cross else x
        q = self.head_to_batch_dim(self.to_q(x))
        k = self.head_to_batch_dim(self.to_k(ctx))
        v = self.head_to_batch_dim(self.to_v(ctx))
        out = None
        if self.ctrl is not None:
            out = self.ctrl(is_cross, self.layer_idx, self.place, q, k, v, self.scale, self.heads)
        if out is None:
            attn = (torch.einsum("bid,bjd->bij", q, k) * self.scale).softmax(dim=-1)
            out = self.batch_to_head_dim(torch.einsum("bij,bjd->bid", attn, v))
        return self.to_out[0](out)


class GEGLU(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.proj = nn.Linear(cin, cout * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(c, 4 * c), nn.Identity(), nn.Linear(4 * c, c)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, c, cross_dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(c)
        self.attn1 = Attention(c, None, heads)
        self.norm2 = nn.LayerNorm(c)
        self.attn2 = Attention(c, cross_dim, heads)
        self.norm3 = nn.LayerNorm(c)
        self.ff = FeedForward(c)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), encoder_hidden_states=ctx)
        x = x + self.ff(self.norm3(x))
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, c, cross_dim, heads, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.proj_in = nn.Conv2d(c, c, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(c, cross_dim, heads)])
        self.proj_out = nn.Conv2d(c, c, 1)

    def forward(self, x, ctx):
        b, c, hh, ww = x.shape
        res = x
        h = self.proj_in(self.norm(x))
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = self.transformer_blocks[0](h, ctx)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        return self.proj_out(h) + res


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, cross_dim, heads, has_attn, add_down, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups=groups) for i in range(2)])
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, cross_dim, heads, groups) for _ in range(2)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, cross_dim, heads, groups):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, cross_dim, heads, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups=groups), ResnetBlock2D(c, c, groups=groups)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev_out, cross_dim, heads, has_attn, add_up, groups):
        super().__init__()
        res = []
        for i in range(3):
            skip = cin if i == 2 else cout
            rin = prev_out if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, groups=groups))
        self.resnets = nn.ModuleList(res)
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, cross_dim, heads, groups) for _ in range(3)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNet2DConditionModel(nn.Module):
    """SD1.x: block_out_channels (320,640,1280,1280), 2 layers/block, 8 heads, ctx dim 768."""

    def __init__(self, block_out_channels=(320, 640, 1280, 1280), cross_dim=768, heads=8, groups=32,
                 in_ch=4, out_ch=4):
        super().__init__()
        ch = block_out_channels
        self.ch0 = ch[0]
        temb_ch = ch[0] * 4
        self.conv_in = nn.Conv2d(in_ch, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb_ch)
        downs, cin = [], ch[0]
        for i, cout in enumerate(ch):
            downs.append(DownBlock(cin, cout, cross_dim, heads, has_attn=(i < 3), add_down=(i < len(ch) - 1), groups=groups))
            cin = cout
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(ch[-1], cross_dim, heads, groups)
        rev = list(reversed(ch))
        ups, prev = [], rev[0]
        for i, cout in enumerate(rev):
            cin_skip = rev[min(i + 1, len(ch) - 1)]
            ups.append(UpBlock(cin_skip, cout, prev, cross_dim, heads, has_attn=(i > 0), add_up=(i < len(ch) - 1), groups=groups))
            prev = cout
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(groups, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], out_ch, 3, padding=1)
        if temb_ch != 1280:
            # toy widths (tests only): time_emb_proj input follows 4*ch[0]
            for m in self.modules():
                if isinstance(m, ResnetBlock2D):
                    m.time_emb_proj = nn.Linear(temb_ch, m.conv1.out_channels)
        self._index_attention()

    def _index_attention(self):
        idx = 0
        for name, child in self.named_children():
            place = "down" if "down" in name else "up" if "up" in name else "mid" if "mid" in name else None
            if place is None:
                continue
            for m in child.modules():
                if isinstance(m, Attention):
                    m.layer_idx, m.place = idx, place
                    idx += 1
        self.num_attention = idx

    def attention_modules(self):
        return [m for m in self.modules() if isinstance(m, Attention)]

    def set_ctrl(self, ctrl):
        for m in self.attention_modules():
            m.ctrl = ctrl

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states=None):
        b = sample.shape[0]
        t = torch.as_tensor(timestep)
        if t.dim() == 0:
            t = t[None].expand(b)
        temb = self.time_embedding(timestep_embedding(t, self.ch0).to(sample.dtype))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips.extend(outs)
        x = self.mid_block(x, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return {"sample": x}


def synthetic_tensor(name: str, shape, seed: int = 0, res_gain: float = 0.5) -> torch.Tensor:
    """Deterministic synthetic value of one SD1.x parameter.  Seeded per NAME
    (crc32(name) + 1000003*seed) so the result does not depend on enumeration order; the product
    side (`etainv/weights.py`) implements the same rule independently.
      1-D `*.weight` (GroupNorm/LayerNorm scale): 1 + 0.1 N(0,1);   other 1-D (biases): 0.05 N(0,1);
      matrices / conv kernels: N(0,1)/sqrt(fan_in), x res_gain on residual-branch outputs
      (conv2, to_out, ff.net.2, proj_out) to keep the random network well-conditioned in fp16."""
    import zlib
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    if len(shape) == 1:
        if name.endswith(".weight"):
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    gain = res_gain if (name.endswith("conv2.weight") or ".to_out." in name or ".ff.net.2." in name
                        or ".proj_out." in name) else 1.0
    return torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))


@torch.no_grad()
def init_synthetic_(unet: nn.Module, seed: int = 0) -> nn.Module:
    for name, p in unet.named_parameters():
        p.copy_(synthetic_tensor(name, p.shape, seed))
    return unet


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())


def build_unet(seed: int = 0, block_out_channels=(320, 640, 1280, 1280), **kw) -> UNet2DConditionModel:
    """Construct on the meta device (skips torch's default init of 860 M parameters), materialise
    on CPU and fill with the deterministic synthetic weights."""
    with torch.device("meta"):
        u = UNet2DConditionModel(block_out_channels=block_out_channels, **kw)
    u = u.to_empty(device="cpu")
    u._index_attention()
    init_synthetic_(u, seed)
    return u.eval()
