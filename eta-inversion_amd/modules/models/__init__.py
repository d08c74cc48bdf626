"""Pipeline object + image pre/post-processing (reference modules/models/__init__.py:12-138).

`load_diffusion_model` returns `(pipeline, (preproc, postproc))` like the reference.  The pipeline's `.unet` is the native
MI355X engine; `.vae` and `.text_encoder` (third-party networks outside the DDIM loop) run on the same
kernels (etainv/nets.py).  `ETAINV_SD_PATH=<diffusers snapshot>` supplies real weights for all three; without it every
network gets seeded synthetic weights (there is no network access to fetch checkpoints)."""
import os
from pathlib import Path

import numpy as np
import torch

from etainv.engine import Engine
from etainv.nets import NativeCLIPText, NativeVAE
from etainv.weights import load_component
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


class EtaPipeline:
    def __init__(self, device="cuda", dtype=torch.float16, latent_size=64, max_img=1, seed=0):
        self.device = torch.device(device if device != "cuda" else "cuda:0")
        self.engine = Engine(dtype=dtype, max_unet_batch=4 * max_img, latent_size=latent_size, max_img=max_img, device=str(self.device))
        self.engine.load_default(seed)
        self.unet = NativeUNet(self.engine, torch.float32)
        self.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                                       set_alpha_to_one=False)
        self.tokenizer = load_tokenizer()
        sd = os.environ.get("ETAINV_SD_PATH")
        self.text_encoder = NativeCLIPText(load_component(sd, "text_encoder") if sd else None, dtype, seed)
        self.vae = NativeVAE(load_component(sd, "vae") if sd else None, dtype, seed)


class StablePreprocess:
    """image file / uint8 array -> (1,3,size,size) float32 in [-1,1] (reference :12-76).  cv2 is not in this image: the file is decoded
    with PIL (RGB, what cv2.imread + COLOR_BGR2RGB yields) and `cv2.resize` (INTER_LINEAR on uint8, no antialiasing) is restated in
    modules/models/resize.py; `pil_resize=True` is PIL's default resize filter (bicubic) like the reference."""

    def __init__(self, device, size=512, return_np=False, center_crop=False, pil_resize=False):
        self.device, self.size, self.return_np, self.center_crop, self.pil_resize = device, size, return_np, center_crop, pil_resize

    def __call__(self, image):
        from PIL import Image
        if isinstance(image, (str, Path)):
            image = np.array(Image.open(str(image)).convert("RGB"))
        if self.center_crop:                          # the centred square of side min(h, w) (reference :45-61)
            h, w = image.shape[:2]
            side = min(h, w)
            top, left = (h - side) // 2, (w - side) // 2
            image = image[top:top + side, left:left + side]
        if self.pil_resize:
            image = np.array(Image.fromarray(image).resize((self.size, self.size)))
        else:
            from .resize import resize_linear_u8
            image = resize_linear_u8(np.ascontiguousarray(image), (self.size, self.size))
        image_pt = (torch.from_numpy(image).float() / 127.5 - 1).permute(2, 0, 1).unsqueeze(0).to(self.device)
        return (image_pt, image) if self.return_np else image_pt


class StablePostProc:
    def __call__(self, image):
        image = (image / 2 + 0.5).clamp(0, 1)
        return (image.float().cpu().permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)[0]


def load_diffusion_model(model="CompVis/stable-diffusion-v1-4", device="cuda", preproc_args=None, variant=None, **kwargs):
    """reference modules/models/__init__.py:100-138.  Precision (`variant`, the reference's `--prec`): None / "fp32" -- the reference's default
    (`edit_image.py:147` without `--prec`, `modules/models/__init__.py:104-138`) -- runs the engine with fp32 operands on the f32-input matrix
    instruction (csrc/f32path.hip; about 1/16 of the 16-bit rate: the parity mode in which the edited latents meet rtol 1e-3 / atol 1e-4 against
    the fp32 reference path); "fp16" / "bf16" run 16-bit MFMA operands with fp32 accumulation (latents, contexts, scheduler state and the attention
    softmax stay fp32) -- the throughput modes; eval.py and bench.py pass their precision explicitly."""
    if variant is None:
        variant = "fp32"
    print(f"Loading model {model} ({variant}) ...")
    if model not in ("sd14", "sd15", "CompVis/stable-diffusion-v1-4", "runwayml/stable-diffusion-v1-5"):
        raise Exception(model)
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[variant]
    pipe = EtaPipeline(device=device, dtype=dtype, **kwargs)
    return pipe, (StablePreprocess(pipe.device, size=8 * pipe.engine.L, **(preproc_args or {})), StablePostProc())
