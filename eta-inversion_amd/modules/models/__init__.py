"""Pipeline object + image pre/post-processing (reference modules/models/__init__.py:12-138).

`load_diffusion_model` returns `(pipeline, (preproc, postproc))` like the reference.  The pipeline's `.unet` is the native
MI355X engine.  The VAE and the CLIP text encoder are third-party networks outside the DDIM loop ("next"):
when `ETAINV_SD_PATH` is not set, deterministic stand-ins are used so the plumbing (CLI, editors) runs end to end."""
import os
import zlib
from pathlib import Path

import numpy as np
import torch

from etainv.engine import Engine
from ..schedulers import DDIMScheduler
from ..utils.tokenizer import load_tokenizer


class NativeUNet:
    """`model.unet` of the reference: callable `(latent, t, encoder_hidden_states=ctx) -> {"sample": eps}` backed by
    etainv_unet_forward.  `attn_ctrl` is the declarative attention control for the next calls (set by controllers)."""

    def __init__(self, engine: Engine, io_dtype):
        self.engine, self.dtype = engine, io_dtype
        self.attn_ctrl = None

    def __call__(self, sample, timestep, encoder_hidden_states=None):
        n = encoder_hidden_states.shape[0]
        assert sample.shape[0] == n, "latent and context batch must match (duplicate the latent for CFG)"
        eps = self.engine.unet(sample.to(self.dtype), timestep, encoder_hidden_states.to(self.dtype), self.attn_ctrl)
        return {"sample": eps}

    forward = __call__


class StandInTextEncoder:
    """Deterministic embedding stand-in for CLIPTextModel: one seeded N(0,1) vector per token id plus a positional term."""
    def __init__(self, device, dim=768):
        self.device, self.dim, self._cache = device, dim, {}

    def _vec(self, key, scale):
        if key not in self._cache:
            g = torch.Generator().manual_seed(zlib.crc32(repr(key).encode()))
            self._cache[key] = scale * torch.randn(self.dim, generator=g)
        return self._cache[key]

    def __call__(self, input_ids):
        ids = input_ids.cpu()
        out = torch.stack([torch.stack([self._vec(("tok", int(t)), 1.0) + self._vec(("pos", p), 0.3) for p, t in enumerate(row)])
                           for row in ids])
        return (out.to(self.device),)


class StandInVAE:
    """Stand-in for AutoencoderKL (outside the hot loop): 8x8 average pooling to 4 channels (RGB + luma) and nearest
    upsampling back; keeps the `(1,3,512,512) <-> (1,4,64,64)` contract of DiffusionInversion.encode/decode."""
    dtype = torch.float32

    def encode(self, image):
        x = torch.nn.functional.avg_pool2d(image.float(), 8)
        z = torch.cat([x, x.mean(1, keepdim=True)], 1) / 0.18215 * 0.5

        class _D:
            mean = z
        return {"latent_dist": _D()}

    def decode(self, z):
        return {"sample": torch.nn.functional.interpolate(z[:, :3].float() * 0.18215 * 2.0, scale_factor=8.0, mode="nearest")}


class EtaPipeline:
    def __init__(self, device="cuda", dtype=torch.float16, latent_size=64, max_img=1, seed=0):
        self.device = torch.device(device if device != "cuda" else "cuda:0")
        self.engine = Engine(dtype=dtype, max_unet_batch=4 * max_img, latent_size=latent_size, max_img=max_img, device=str(self.device))
        self.engine.load_default(seed)
        self.unet = NativeUNet(self.engine, torch.float32)
        self.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                       set_alpha_to_one=False)
        self.tokenizer = load_tokenizer()
        self.text_encoder = StandInTextEncoder(self.device)
        self.vae = StandInVAE()
        sd = os.environ.get("ETAINV_SD_PATH")
        if sd and os.path.isdir(os.path.join(sd, "text_encoder")):
            from transformers import CLIPTextModel
            enc = CLIPTextModel.from_pretrained(os.path.join(sd, "text_encoder")).to(self.device).eval()
            self.text_encoder = lambda ids: enc(ids)


class StablePreprocess:
    """image file / uint8 array -> (1,3,size,size) float32 in [-1,1] (reference :12-76; PIL replaces cv2, which is not in
    the image: bilinear resize, RGB order)."""

    def __init__(self, device, size=512, return_np=False, center_crop=False, pil_resize=False):
        self.device, self.size, self.return_np, self.center_crop, self.pil_resize = device, size, return_np, center_crop, pil_resize

    def __call__(self, image):
        from PIL import Image
        if isinstance(image, (str, Path)):
            image = np.array(Image.open(str(image)).convert("RGB"))
        if self.center_crop:
            h, w = image.shape[:2]
            if w > h:
                x1 = (w - h) // 2
                x2 = w - h - x1
                if x2 > 0:
                    image = image[:, x1:-x2]
            else:
                y1 = (h - w) // 2
                y2 = h - w - y1
                if y2 > 0:
                    image = image[y1:-y2]
        resample = Image.BICUBIC if self.pil_resize else Image.BILINEAR
        image = np.array(Image.fromarray(image).resize((self.size, self.size), resample))
        image_pt = (torch.from_numpy(image).float() / 127.5 - 1).permute(2, 0, 1).unsqueeze(0).to(self.device)
        return (image_pt, image) if self.return_np else image_pt


class StablePostProc:
    def __call__(self, image):
        image = (image / 2 + 0.5).clamp(0, 1)
        return (image.float().cpu().permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)[0]


def load_diffusion_model(model="CompVis/stable-diffusion-v1-4", device="cuda", preproc_args=None, variant=None, **kwargs):
    variant = variant or "fp32"
    print(f"Loading model {model} ({variant}) ...")
    if model not in ("sd14", "sd15", "CompVis/stable-diffusion-v1-4", "runwayml/stable-diffusion-v1-5"):
        raise Exception(model)
    # fp32 is not an MFMA operand type: "fp32" keeps fp32 latents/contexts at the boundary with fp16 operands + fp32 accumulate
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float16}[variant]
    pipe = EtaPipeline(device=device, dtype=dtype, **kwargs)
    return pipe, (StablePreprocess(pipe.device, size=8 * pipe.engine.L, **(preproc_args or {})), StablePostProc())
