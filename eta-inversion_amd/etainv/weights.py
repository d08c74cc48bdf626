"""UNet weights for the engine: deterministic synthetic SD1.x-shaped parameters (no network / no checkpoint in the
build image) or a local diffusers snapshot (`unet/diffusion_pytorch_model.safetensors`)."""
import math
import zlib
from pathlib import Path

import torch

_RES_OUT = ("conv2.weight", ".to_out.", ".ff.net.2.", ".proj_out.")


def synthetic_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Synthetic value of one parameter, a pure function of (name, shape, seed):
    norm scales 1 + 0.1 N(0,1); biases 0.05 N(0,1); matrices / kernels N(0,1)/sqrt(fan_in), halved on the
    residual-branch output layers so the random network stays well inside fp16 range."""
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1:
        base = torch.randn(shape, generator=g)
        return 1.0 + 0.1 * base if name.endswith(".weight") else 0.05 * base
    fan_in = math.prod(shape[1:])
    gain = 0.5 if (name.endswith(_RES_OUT[0]) or any(k in name for k in _RES_OUT[1:])) else 1.0
    return torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))


def load_snapshot(path) -> dict:
    """state dict of `<path>/unet/diffusion_pytorch_model(.fp16).safetensors` (diffusers layout)."""
    from safetensors.torch import load_file
    root = Path(path)
    for cand in ("unet/diffusion_pytorch_model.safetensors", "unet/diffusion_pytorch_model.fp16.safetensors",
                 "diffusion_pytorch_model.safetensors"):
        if (root / cand).exists():
            return {k: v.float() for k, v in load_file(str(root / cand)).items()}
    raise FileNotFoundError(f"no UNet safetensors under {root}")
