"""SD1.x AutoencoderKL restated for the CPU oracle (test infrastructure).  Third-party network ([3P] diffusers 0.21.1
`AutoencoderKL`, not vendored in the reference): called at modules/inversion/diffusion_inversion.py:193 (`vae.decode(z /
0.18215)["sample"]`) and :206 (`vae.encode(img)["latent_dist"].mean * 0.18215`).  Published architecture: block_out_channels
(128,256,512,512), 2 layers per block, GroupNorm(32, eps 1e-6), SiLU, single-head attention in the mid block, asymmetric
(0,1,0,1) padding in the stride-2 downsamplers, 8-channel moments + 1x1 quant / post_quant convs.  Parameter names equal the
diffusers state-dict keys (`to_q/to_k/to_v/to_out.0` attention naming).  Numerics vs diffusers: parity unpinned."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Res(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if self.conv_shortcut is not None else x) + h


class Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        a = (q @ k.transpose(1, 2) * c ** -0.5).softmax(-1)
        o = self.to_out[0](a @ v)
        return x + o.transpose(1, 2).reshape(b, c, h, w)


class Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([Attn(c)])
        self.resnets = nn.ModuleList([Res(c, c), Res(c, c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Conv(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=0 if stride == 2 else 1)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, down):
        super().__init__()
        self.resnets = nn.ModuleList([Res(cin, cout), Res(cout, cout)])
        self.downsamplers = nn.ModuleList([_Conv(cout, 2)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1)))
        return x


class UpBlock(nn.Module):
    def __init__(self, cin, cout, up):
        super().__init__()
        self.resnets = nn.ModuleList([Res(cin, cout), Res(cout, cout), Res(cout, cout)])
        self.upsamplers = nn.ModuleList([_Conv(cout, 1)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class Encoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512)):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        blocks, cin = [], ch[0]
        for i, c in enumerate(ch):
            blocks.append(DownBlock(cin, c, i < len(ch) - 1))
            cin = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = Mid(ch[-1])
        self.conv_norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 8, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512)):
        super().__init__()
        rev = list(reversed(ch))
        self.conv_in = nn.Conv2d(4, rev[0], 3, padding=1)
        self.mid_block = Mid(rev[0])
        blocks, cin = [], rev[0]
        for i, c in enumerate(rev):
            blocks.append(UpBlock(cin, c, i < len(ch) - 1))
            cin = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder, self.decoder = Encoder(), Decoder()
        self.quant_conv = nn.Conv2d(8, 8, 1)
        self.post_quant_conv = nn.Conv2d(4, 4, 1)

    def encode_mean(self, img):
        return self.quant_conv(self.encoder(img))[:, :4]

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def build_vae(seed=0):
    from .unet import synthetic_tensor
    with torch.device("meta"):
        m = AutoencoderKL()
    m = m.to_empty(device="cpu")
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(synthetic_tensor("vae." + name, p.shape, seed))
    return m.eval()
