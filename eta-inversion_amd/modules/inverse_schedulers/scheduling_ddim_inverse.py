"""Inverse DDIM scheduler (reference modules/inverse_schedulers/scheduling_ddim_inverse.py:10-142): same constructor,
`from_scheduler`, `set_timesteps`, `timesteps`, `step -> Output(prev_sample)`, `inv_steps` in {sameshift, samesame,
shiftshift}.  The arithmetic of `ddim_step` runs in the HIP kernel etainv_ddim_step on the latent's device."""
from collections import namedtuple

import torch

from etainv import _capi
from ..schedulers import DDIMScheduler
from .diffusion_inverse_scheduler import DiffusionInverseScheduler


class DDIMInverseScheduler(DiffusionInverseScheduler):
    Output = namedtuple("DDIMInverseSchedulerOutput", ("prev_sample",))

    def __init__(self, scheduler: DDIMScheduler, inv_steps: str = "sameshift") -> None:
        self.scheduler = scheduler
        self.is_backward = False
        self.inv_steps = inv_steps

    @staticmethod
    def from_scheduler(scheduler, inv_steps: str = "sameshift", **kwargs) -> "DDIMInverseScheduler":
        return DDIMInverseScheduler(DDIMScheduler.from_config({**scheduler.config, **kwargs}), inv_steps=inv_steps)

    def set_timesteps(self, num_inference_steps: int) -> None:
        self.scheduler.set_timesteps(num_inference_steps)

    @property
    def timesteps(self):
        steps = torch.flip(self.scheduler.timesteps, dims=(0,))
        if self.scheduler.config.steps_offset != 0:
            assert steps[0] == 1
        if self.inv_steps == "shiftshift":
            steps = torch.stack([self.get_timestep(s, -1) for s in steps])
        return steps

    def get_timestep(self, timestep, offset: int):
        return timestep + offset * (self.scheduler.config.num_train_timesteps // self.scheduler.num_inference_steps)

    def _alpha(self, t) -> float:
        t = min(int(t), 999)                                   # clamp at the training horizon
        return float(self.scheduler.alphas_cumprod[t]) if t >= 0 else float(self.scheduler.final_alpha_cumprod)

    def ddim_step(self, sample, model_output, timestep_from, timestep_to):
        x, eps = sample.contiguous(), model_output.contiguous()
        out = torch.empty_like(x)
        _capi.check(_capi.load().etainv_ddim_step(_capi.ptr(x), _capi.ptr(eps), self._alpha(timestep_from), self._alpha(timestep_to),
                                                 _capi.ptr(out), x.numel(), _capi.dtype_code(x.dtype), _capi.stream_ptr()))
        return out

    def step(self, noise_pred, t, latent) -> "DDIMInverseScheduler.Output":
        if not self.is_backward:
            if self.inv_steps == "sameshift":
                t_from, t_to = self.get_timestep(t, -1), t
            elif self.inv_steps in ("samesame", "shiftshift"):
                t_from, t_to = t, self.get_timestep(t, +1)
            else:
                raise Exception(self.inv_steps)
        else:
            t_from, t_to = t, self.get_timestep(t, -1)
        return DDIMInverseScheduler.Output(self.ddim_step(latent, noise_pred, t_from, t_to))
