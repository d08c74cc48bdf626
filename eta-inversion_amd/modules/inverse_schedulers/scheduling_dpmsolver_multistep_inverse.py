"""DPM-Solver++ inverse scheduler (reference modules/inverse_schedulers/scheduling_dpmsolver_multistep_inverse.py:10-159): same constructor,
`from_scheduler`, `set_timesteps`, `timesteps`, `get_first_neg_step`, `step -> Output(prev_sample)`, `inv_steps` in {samesame, sameshift,
shiftshift}.  The reference wraps [3P] diffusers' DPMSolverMultistepInverseScheduler; here the wrapped object is the restated solver of
modules/schedulers.py run on the ascending timestep grid (0 first, the step after the last lands on the noisiest timestep 999)."""
from collections import namedtuple
from typing import Any, Dict

import numpy as np
import torch

from ..schedulers import DPMSolverMultistepScheduler
from .diffusion_inverse_scheduler import DiffusionInverseScheduler


class _InverseCore(DPMSolverMultistepScheduler):
    """[3P] diffusers DPMSolverMultistepInverseScheduler.set_timesteps: the solver's grid ascending, noisiest_timestep = 999"""

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.noisiest_timestep = self.config.num_train_timesteps - 1
        self.timesteps = torch.from_numpy(self._grid(num_inference_steps)[:-1].copy().astype(np.int64))
        self.model_outputs, self.lower_order_nums = [None] * self.config.solver_order, 0


class DPMSolverMultistepInverseScheduler(DiffusionInverseScheduler):
    Output = namedtuple("DPMSolverMultistepInverseSchedulerOutput", ("prev_sample",))

    def __init__(self, cfg: Dict[str, Any], inv_steps: str = "samesame") -> None:
        assert inv_steps in ("samesame", "sameshift", "shiftshift")
        self.sched = _InverseCore.from_config(cfg)
        self.inv_steps = inv_steps

    @staticmethod
    def from_scheduler(scheduler, inv_steps: str = "samesame", **kwargs) -> "DPMSolverMultistepInverseScheduler":
        return DPMSolverMultistepInverseScheduler({**scheduler.config, **kwargs}, inv_steps=inv_steps)

    def set_timesteps(self, num_inference_steps: int) -> None:
        self.sched.set_timesteps(num_inference_steps)
        steps = self.sched.timesteps
        assert steps[0] == 0
        if self.inv_steps == "shiftshift":
            steps = torch.cat([torch.as_tensor(self.get_first_neg_step())[None], steps[:-1]])
        self.sched.timesteps = steps

    @property
    def timesteps(self) -> torch.Tensor:
        return self.sched.timesteps

    def get_first_neg_step(self):
        return self.sched.timesteps[0] - (self.sched.timesteps[1] - self.sched.timesteps[0])

    def step(self, noise_pred, t, latent) -> "DPMSolverMultistepInverseScheduler.Output":
        """reference :83-159.  Negative timesteps (first step of "sameshift" / "shiftshift") index the tables from the end, like the
        tensors of the reference do."""
        s = self.sched
        if s.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        idx = (s.timesteps == int(t)).nonzero()
        step_index = len(s.timesteps) - 1 if len(idx) == 0 else int(idx[0])
        timestep = int(t)
        if self.inv_steps == "sameshift":
            step_index -= 1
            timestep = int(s.timesteps[step_index]) if step_index >= 0 else int(self.get_first_neg_step())
        n = len(s.timesteps)
        prev_timestep = s.noisiest_timestep if step_index == n - 1 else int(s.timesteps[step_index + 1])
        lower_order_final = step_index == n - 1 and s.config.lower_order_final and n < 15
        x0 = s.convert_model_output(noise_pred, timestep, latent)
        s.model_outputs = s.model_outputs[1:] + [x0]
        if s.lower_order_nums < 1 or lower_order_final:
            out = s.dpm_solver_first_order_update(x0, timestep, prev_timestep, latent)
        else:
            out = s.multistep_dpm_solver_second_order_update(s.model_outputs, [int(s.timesteps[step_index - 1]), timestep], prev_timestep, latent)
        if s.lower_order_nums < s.config.solver_order:
            s.lower_order_nums += 1
        return DPMSolverMultistepInverseScheduler.Output(prev_sample=out)
