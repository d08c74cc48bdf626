"""Backward DDIM scheduler of the pipeline ([3P] diffusers `DDIMScheduler` in the reference; closed form).
Only what the reference's path touches is provided: config, alphas_cumprod, final_alpha_cumprod, timesteps,
set_timesteps, _get_variance, step(eta, variance_noise) -- `step` runs the HIP kernel etainv_ddim_eta_step."""
from collections import namedtuple

import numpy as np
import torch

from etainv import _capi


class _Config(dict):
    __getattr__ = dict.__getitem__


class DDIMScheduler:
    Output = namedtuple("DDIMSchedulerOutput", ("prev_sample",))

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=False, steps_offset=0, prediction_type="epsilon", **extra):
        if beta_schedule != "scaled_linear" or prediction_type != "epsilon" or clip_sample:
            raise NotImplementedError("only the SD1.x configuration (scaled_linear, epsilon, no clipping) is built")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                              steps_offset=steps_offset, prediction_type=prediction_type, **extra)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @classmethod
    def from_config(cls, config):
        return cls(**dict(config))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def _alpha(self, t):
        return float(self.alphas_cumprod[int(t)]) if int(t) >= 0 else float(self.final_alpha_cumprod)

    def _get_variance(self, timestep, prev_timestep):
        a_t, a_p = self._alpha(timestep), self._alpha(prev_timestep)
        return (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)

    def step(self, model_output, timestep, sample, eta=0.0, variance_noise=None, eta_mask=None, **unused):
        """eta: float; eta_mask: optional (n,H,W) per-pixel multiplier (the reference smuggles a tensor-valued eta through
        an `EtaTensor` subclass, eta_inversion.py:23-33,245 -- here it is an explicit argument)."""
        t = int(timestep)
        p = t - self.config.num_train_timesteps // self.num_inference_steps
        x, eps = sample.contiguous(), model_output.contiguous()
        out = torch.empty_like(x)
        rows, c, hw = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        noise = None
        if variance_noise is not None and float(eta) > 0:
            noise = variance_noise.to(x.dtype).reshape(-1)[: c * hw].contiguous()
        mask = None if eta_mask is None else eta_mask.to(x.dtype).contiguous()
        _capi.check(_capi.load().etainv_ddim_eta_step(_capi.ptr(x), _capi.ptr(eps), float(eta), _capi.ptr(mask),
                                                     0 if mask is None else mask.shape[0], _capi.ptr(noise), self._alpha(t),
                                                     self._alpha(p), self._get_variance(t, p), rows, c, hw, _capi.ptr(out),
                                                     _capi.dtype_code(x.dtype), _capi.stream_ptr()))
        return DDIMScheduler.Output(out)
