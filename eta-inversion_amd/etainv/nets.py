"""SD1.x VAE (AutoencoderKL) and CLIP ViT-L/14 text encoder on the native kernels.

These third-party networks sit outside the DDIM loop (1 encode + 2 decodes + 4 text forwards per image, about 1.5 % of the
FLOPs): the host code below only sequences C-ABI kernel calls (`etainv_op_*`: MFMA implicit-GEMM convs / linears,
GroupNorm, LayerNorm, im2col, row softmax, causal attention) on device buffers.  They replace
`model.vae.encode(...)["latent_dist"].mean`, `model.vae.decode(...)["sample"]` and `model.text_encoder(ids)[0]`
(reference modules/inversion/diffusion_inversion.py:193, 206, 230).  Weights: diffusers / transformers state-dict names."""
import math
import zlib

import torch

from . import _capi
from .weights import synthetic_tensor


def _dev(t, dtype):
    return t.detach().to(device="cuda", dtype=dtype).contiguous()


class _Ops:
    def __init__(self, dtype):
        self.lib, self.dtype, self.code = _capi.load(), dtype, _capi.dtype_code(dtype)
        self._scratch = {}

    def st(self):
        return _capi.stream_ptr()

    def gemm(self, a, w, bias=None, residual=None):
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty(m, n, dtype=self.dtype, device=a.device)
        _capi.check(self.lib.etainv_op_gemm(_capi.ptr(a), _capi.ptr(w), _capi.ptr(bias), _capi.ptr(residual), _capi.ptr(out), m, n, k, 0,
                                            self.code, self.st()))
        return out

    def conv3(self, x, w, bias, stride=1, ups=0, pad0=0, residual=None, out_nchw=0):
        b, h, wd, cin = x.shape
        cout = w.shape[0]
        ho = h * 2 if ups else (h // 2 if stride == 2 else h)
        wo = wd * 2 if ups else (wd // 2 if stride == 2 else wd)
        if out_nchw:
            out = torch.empty(b, out_nchw, ho, wo, dtype=torch.float32, device=x.device)
        else:
            out = torch.empty(b, ho, wo, cout, dtype=self.dtype, device=x.device)
        _capi.check(self.lib.etainv_op_conv3x3_ex(_capi.ptr(x), _capi.ptr(w), _capi.ptr(bias), _capi.ptr(residual), _capi.ptr(out), b, h, wd,
                                                  cin, cout, stride, ups, pad0, out_nchw, _capi.F32, self.code, self.st()))
        return out

    def gn(self, x, gamma, beta, silu, eps=1e-6):
        b, c = x.shape[0], x.shape[-1]
        hw = x.numel() // (b * c)
        out = torch.empty_like(x)
        key = ("gn", b)
        if key not in self._scratch:
            self._scratch[key] = torch.zeros(b * 65 * 64, dtype=torch.float32, device=x.device)
        _capi.check(self.lib.etainv_op_groupnorm(_capi.ptr(x), None, c, 0, _capi.ptr(gamma), _capi.ptr(beta), _capi.ptr(out), b, hw, 32, eps,
                                                 int(silu), _capi.ptr(self._scratch[key]), self.code, self.st()))
        return out

    def ln(self, x, gamma, beta, eps=1e-5):
        rows, c = x.shape
        out = torch.empty_like(x)
        _capi.check(self.lib.etainv_op_layernorm(_capi.ptr(x), _capi.ptr(gamma), _capi.ptr(beta), _capi.ptr(out), rows, c, eps, self.code, self.st()))
        return out

    def im2col(self, x_nchw, premix=None):
        b, cin, h, w = x_nchw.shape
        x_nchw = x_nchw.float().contiguous()
        out = torch.empty(b * h * w, 64, dtype=self.dtype, device=x_nchw.device)
        _capi.check(self.lib.etainv_op_im2col3x3(_capi.ptr(x_nchw), _capi.F32, cin, h, w, b, _capi.ptr(premix), _capi.ptr(out), self.code, self.st()))
        return out


def _pack_conv3(w):                       # [O][I][3][3] -> [O][9][I]
    return w.permute(0, 2, 3, 1).contiguous()


def _pack_conv_small(w):                  # [O][cin<=4][3][3] -> [O][64], k = tap*cin + ci
    o, cin = w.shape[:2]
    out = torch.zeros(o, 64)
    out[:, : 9 * cin] = w.permute(0, 2, 3, 1).reshape(o, 9 * cin)
    return out


class NativeVAE:
    dtype = torch.float32   # boundary dtype (images / latents are fp32 at the boundary)

    def __init__(self, state_dict=None, compute_dtype=torch.float16, seed=0):
        self.ops = _Ops(compute_dtype)
        self.cd = compute_dtype
        self._sd, self._seed = state_dict, seed
        self.w = {}
        self._build()

    def _get(self, name, shape):
        if self._sd is not None:
            t = self._sd[name].float()
            assert tuple(t.shape) == tuple(shape), (name, t.shape, shape)
            return t
        return synthetic_tensor("vae." + name, shape, self._seed)

    def _res(self, prefix, cin, cout):
        g = self._get
        d = {"n1": (_dev(g(prefix + ".norm1.weight", (cin,)), torch.float32), _dev(g(prefix + ".norm1.bias", (cin,)), torch.float32)),
             "c1": (_dev(_pack_conv3(g(prefix + ".conv1.weight", (cout, cin, 3, 3))), self.cd), _dev(g(prefix + ".conv1.bias", (cout,)), torch.float32)),
             "n2": (_dev(g(prefix + ".norm2.weight", (cout,)), torch.float32), _dev(g(prefix + ".norm2.bias", (cout,)), torch.float32)),
             "c2": (_dev(_pack_conv3(g(prefix + ".conv2.weight", (cout, cout, 3, 3))), self.cd), _dev(g(prefix + ".conv2.bias", (cout,)), torch.float32)),
             "sc": None}
        if cin != cout:
            d["sc"] = (_dev(g(prefix + ".conv_shortcut.weight", (cout, cin, 1, 1)).reshape(cout, cin), self.cd),
                       _dev(g(prefix + ".conv_shortcut.bias", (cout,)), torch.float32))
        return d

    def _attn(self, prefix, c):
        g = self._get
        lin = lambda n, bias=True: (_dev(g(f"{prefix}.{n}.weight", (c, c)), self.cd), _dev(g(f"{prefix}.{n}.bias", (c,)), torch.float32))
        return {"gn": (_dev(g(prefix + ".group_norm.weight", (c,)), torch.float32), _dev(g(prefix + ".group_norm.bias", (c,)), torch.float32)),
                "q": lin("to_q"), "k": lin("to_k"), "v": lin("to_v"), "o": lin("to_out.0")}

    def _conv(self, prefix, c):
        return (_dev(_pack_conv3(self._get(prefix + ".weight", (c, c, 3, 3))), self.cd), _dev(self._get(prefix + ".bias", (c,)), torch.float32))

    def _build(self):
        g, ch = self._get, (128, 256, 512, 512)
        e, d = {}, {}
        e["conv_in"] = (_dev(_pack_conv_small(g("encoder.conv_in.weight", (128, 3, 3, 3))), self.cd), _dev(g("encoder.conv_in.bias", (128,)), torch.float32))
        cin = ch[0]
        e["down"] = []
        for i, c in enumerate(ch):
            blk = {"res": [self._res(f"encoder.down_blocks.{i}.resnets.0", cin, c), self._res(f"encoder.down_blocks.{i}.resnets.1", c, c)],
                   "down": self._conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c) if i < 3 else None}
            e["down"].append(blk)
            cin = c
        e["mid"] = (self._res("encoder.mid_block.resnets.0", 512, 512), self._attn("encoder.mid_block.attentions.0", 512),
                    self._res("encoder.mid_block.resnets.1", 512, 512))
        e["norm_out"] = (_dev(g("encoder.conv_norm_out.weight", (512,)), torch.float32), _dev(g("encoder.conv_norm_out.bias", (512,)), torch.float32))
        # conv_out (512 -> 8 moments) followed by the 1x1 quant_conv; only the 4 mean channels are needed: fold both
        w_out, b_out = g("encoder.conv_out.weight", (8, 512, 3, 3)), g("encoder.conv_out.bias", (8,))
        q, qb = g("quant_conv.weight", (8, 8, 1, 1)).reshape(8, 8), g("quant_conv.bias", (8,))
        w_mean = torch.einsum("oc,cikl->oikl", q[:4], w_out)
        b_mean = q[:4] @ b_out + qb[:4]
        e["conv_out"] = (_dev(_pack_conv3(w_mean), self.cd), _dev(b_mean, torch.float32))
        # decoder
        pq, pqb = g("post_quant_conv.weight", (4, 4, 1, 1)).reshape(4, 4), g("post_quant_conv.bias", (4,))
        d["premix"] = _dev(torch.cat([pq, pqb[:, None]], 1), torch.float32)
        d["conv_in"] = (_dev(_pack_conv_small(g("decoder.conv_in.weight", (512, 4, 3, 3))), self.cd), _dev(g("decoder.conv_in.bias", (512,)), torch.float32))
        d["mid"] = (self._res("decoder.mid_block.resnets.0", 512, 512), self._attn("decoder.mid_block.attentions.0", 512),
                    self._res("decoder.mid_block.resnets.1", 512, 512))
        rev, cin = (512, 512, 256, 128), 512
        d["up"] = []
        for i, c in enumerate(rev):
            blk = {"res": [self._res(f"decoder.up_blocks.{i}.resnets.0", cin, c), self._res(f"decoder.up_blocks.{i}.resnets.1", c, c),
                           self._res(f"decoder.up_blocks.{i}.resnets.2", c, c)],
                   "up": self._conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c) if i < 3 else None}
            d["up"].append(blk)
            cin = c
        d["norm_out"] = (_dev(g("decoder.conv_norm_out.weight", (128,)), torch.float32), _dev(g("decoder.conv_norm_out.bias", (128,)), torch.float32))
        w3 = torch.zeros(4, 128, 3, 3)
        w3[:3] = g("decoder.conv_out.weight", (3, 128, 3, 3))
        b3 = torch.zeros(4)
        b3[:3] = g("decoder.conv_out.bias", (3,))
        d["conv_out"] = (_dev(_pack_conv3(w3), self.cd), _dev(b3, torch.float32))
        self.enc, self.dec = e, d

    # ------------------------------------------------------------------ blocks
    def _run_res(self, r, x):
        o = self.ops
        h = o.conv3(o.gn(x, *r["n1"], True), *r["c1"])
        res = x
        if r["sc"] is not None:
            b, hh, ww, c = x.shape
            res = o.gemm(x.reshape(b * hh * ww, c), *r["sc"]).reshape(b, hh, ww, -1)
        return o.conv3(o.gn(h, *r["n2"], True), *r["c2"], residual=res)

    def _run_attn(self, a, x):
        o = self.ops
        b, hh, ww, c = x.shape
        n = hh * ww
        assert n % 64 == 0, "VAE attention needs H*W to be a multiple of 64"
        t = o.gn(x, *a["gn"], False).reshape(b, n, c)
        q = o.gemm(t.reshape(b * n, c), *a["q"]).reshape(b, n, c)
        k = o.gemm(t.reshape(b * n, c), *a["k"]).reshape(b, n, c)
        outs = []
        for i in range(b):                                                    # scores materialised once per image
            s = o.gemm(q[i], k[i])                                            # [n][n] = Q K^T
            _capi.check(o.lib.etainv_op_row_softmax(_capi.ptr(s), n, n, c ** -0.5, o.code, o.st()))
            vt = o.gemm(a["v"][0], t[i])                                      # [c][n] = (X Wv^T)^T  (bias folded below)
            outs.append(o.gemm(s, vt, a["v"][1]))                             # P V + b_v   (rows of P sum to 1)
        att = torch.stack(outs).reshape(b * n, c)
        return o.gemm(att, *a["o"], residual=x.reshape(b * n, c)).reshape(b, hh, ww, c)

    def _mid(self, m, x):
        return self._run_res(m[2], self._run_attn(m[1], self._run_res(m[0], x)))

    # ------------------------------------------------------------------ API of the reference's `model.vae`
    def encode(self, image):
        """image (B,3,H,W) in [-1,1] -> {"latent_dist": obj with .mean (B,4,H/8,W/8)}"""
        o, e = self.ops, self.enc
        b, _, hh, ww = image.shape
        x = o.gemm(o.im2col(image.cuda()), *e["conv_in"]).reshape(b, hh, ww, 128)
        for blk in e["down"]:
            for r in blk["res"]:
                x = self._run_res(r, x)
            if blk["down"] is not None:
                x = o.conv3(x, *blk["down"], stride=2, pad0=1)
        x = self._mid(e["mid"], x)
        mean = o.conv3(o.gn(x, *e["norm_out"], True), *e["conv_out"], out_nchw=4)

        class _Dist:
            pass
        dist = _Dist()
        dist.mean = mean
        return {"latent_dist": dist}

    def decode(self, z):
        """z (B,4,h,w) -> {"sample": (B,3,8h,8w)}"""
        o, d = self.ops, self.dec
        b, _, hh, ww = z.shape
        x = o.gemm(o.im2col(z.cuda(), d["premix"]), *d["conv_in"]).reshape(b, hh, ww, 512)
        x = self._mid(d["mid"], x)
        for blk in d["up"]:
            for r in blk["res"]:
                x = self._run_res(r, x)
            if blk["up"] is not None:
                x = o.conv3(x, *blk["up"], ups=1)
        return {"sample": o.conv3(o.gn(x, *d["norm_out"], True), *d["conv_out"], out_nchw=3)}


class NativeCLIPText:
    def __init__(self, state_dict=None, compute_dtype=torch.float16, seed=0, layers=12, d=768, heads=12, vocab=49408):
        self.ops, self.cd, self.d, self.heads = _Ops(compute_dtype), compute_dtype, d, heads
        self._sd, self._seed = state_dict, seed
        g = self._get
        pre = "text_model."
        self.tok = _dev(g(pre + "embeddings.token_embedding.weight", (vocab, d)), compute_dtype)
        self.pos = _dev(g(pre + "embeddings.position_embedding.weight", (77, d)), compute_dtype)
        f32 = torch.float32
        self.layers = []
        for i in range(layers):
            p = f"{pre}encoder.layers.{i}."
            wq, wk, wv = (g(p + f"self_attn.{n}.weight", (d, d)) for n in ("q_proj", "k_proj", "v_proj"))
            bq, bk, bv = (g(p + f"self_attn.{n}.bias", (d,)) for n in ("q_proj", "k_proj", "v_proj"))
            self.layers.append({
                "ln1": (_dev(g(p + "layer_norm1.weight", (d,)), f32), _dev(g(p + "layer_norm1.bias", (d,)), f32)),
                "qkv": (_dev(torch.cat([wq, wk, wv]), compute_dtype), _dev(torch.cat([bq, bk, bv]), f32)),
                "out": (_dev(g(p + "self_attn.out_proj.weight", (d, d)), compute_dtype), _dev(g(p + "self_attn.out_proj.bias", (d,)), f32)),
                "ln2": (_dev(g(p + "layer_norm2.weight", (d,)), f32), _dev(g(p + "layer_norm2.bias", (d,)), f32)),
                "fc1": (_dev(g(p + "mlp.fc1.weight", (4 * d, d)), compute_dtype), _dev(g(p + "mlp.fc1.bias", (4 * d,)), f32)),
                "fc2": (_dev(g(p + "mlp.fc2.weight", (d, 4 * d)), compute_dtype), _dev(g(p + "mlp.fc2.bias", (d,)), f32))})
        self.final_ln = (_dev(g(pre + "final_layer_norm.weight", (d,)), f32), _dev(g(pre + "final_layer_norm.bias", (d,)), f32))

    def _get(self, name, shape):
        if self._sd is not None:
            t = self._sd[name].float()
            assert tuple(t.shape) == tuple(shape), (name, t.shape, shape)
            return t
        if "embedding" in name:
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 1000003 * self._seed) & 0x7FFFFFFF)
            return 0.5 * torch.randn(shape, generator=g)
        return synthetic_tensor("clip." + name, shape, self._seed)

    def __call__(self, input_ids):
        """(B,77) int64 -> (last_hidden_state (B,77,768) fp32,)   [`text_encoder(ids)[0]`, diffusion_inversion.py:230]"""
        o, d = self.ops, self.d
        ids = input_ids.to(device="cuda", dtype=torch.int64).contiguous()
        b, n = ids.shape
        x = torch.empty(b * n, d, dtype=self.cd, device="cuda")
        _capi.check(o.lib.etainv_op_embed(_capi.ptr(ids), _capi.ptr(self.tok), _capi.ptr(self.pos), b, n, d, _capi.ptr(x), o.code, o.st()))
        for l in self.layers:
            qkv = o.gemm(o.ln(x, *l["ln1"]), *l["qkv"])
            att = torch.empty(b * n, d, dtype=self.cd, device="cuda")
            _capi.check(o.lib.etainv_op_causal_attention(_capi.ptr(qkv), _capi.ptr(att), b, n, self.heads, d // self.heads, o.code, o.st()))
            x = o.gemm(att, *l["out"], residual=x)
            f = o.gemm(o.ln(x, *l["ln2"]), *l["fc1"])
            _capi.check(o.lib.etainv_op_quick_gelu(_capi.ptr(f), _capi.ptr(f), f.numel(), o.code, o.st()))
            x = o.gemm(f, *l["fc2"], residual=x)
        return (o.ln(x, *self.final_ln).reshape(b, n, d).float(),)
