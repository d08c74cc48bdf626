"""UNet weights for the engine: deterministic synthetic SD1.x-shaped parameters (no network / no checkpoint in the
build image) or a local diffusers snapshot (`unet/diffusion_pytorch_model.safetensors`)."""
import math
import zlib
from pathlib import Path

import torch

_RES_OUT = ("conv2.weight", ".to_out.", ".ff.net.2.", ".proj_out.")


_memo = None          # process-level memo of the synthetic tensors (3.4 GB of host memory for the UNet): off unless asked for


def memoize_synthetic(on: bool = True):
    """Keep every synthetic tensor generated from now on in host memory, so that a process which builds several engines (the test suite,
    A/B benchmarks) pays the 860 M single-threaded `randn` draws once.  Callers must not modify the returned tensors."""
    global _memo
    _memo = {} if on else None


def synthetic_tensor(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Synthetic value of one parameter, a pure function of (name, shape, seed):
    norm scales 1 + 0.1 N(0,1); biases 0.05 N(0,1); matrices / kernels N(0,1)/sqrt(fan_in), halved on the
    residual-branch output layers so the random network stays well inside fp16 range."""
    if _memo is not None:
        key = (name, tuple(int(s) for s in shape), seed)
        if key not in _memo:
            _memo[key] = _synthetic_tensor(name, shape, seed)
        return _memo[key]
    return _synthetic_tensor(name, shape, seed)


def _synthetic_tensor(name, shape, seed):
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


_OLD_VAE_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def load_component(path, sub) -> dict:
    """state dict of `<path>/<sub>/*.safetensors` for sub in {"vae", "text_encoder"}; pre-0.18 diffusers VAE attention
    names (query/key/value/proj_attn) are mapped to to_q/to_k/to_v/to_out.0."""
    from safetensors.torch import load_file
    root = Path(path) / sub
    for cand in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors", "model.fp16.safetensors"):
        if (root / cand).exists():
            sd = {}
            for k, v in load_file(str(root / cand)).items():
                parts = k.split(".")
                if sub == "vae" and "attentions" in parts:
                    k = ".".join(parts[:-2] + [_OLD_VAE_ATTN.get(parts[-2], parts[-2]), parts[-1]]) if parts[-2] in _OLD_VAE_ATTN else k
                v = v.float()
                if sub == "vae" and v.dim() == 4 and ".attentions." in k:     # old checkpoints keep 1x1 convs for q/k/v/out
                    v = v.reshape(v.shape[0], v.shape[1])
                sd[k] = v
            return sd
    raise FileNotFoundError(f"no safetensors under {root}")
