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


class DPMSolverMultistepScheduler:
    """Backward DPM-Solver++(2M) scheduler ([3P] diffusers `DPMSolverMultistepScheduler` in the reference, built by
    DiffusionInversion.create_schedulers for `--scheduler dpm`, modules/inversion/diffusion_inversion.py:139-146).  Restated from the
    published method (arXiv:2211.01095, eqs. 11-13) for the configuration the reference ends up with: solver_order 2, "dpmsolver++",
    midpoint, epsilon prediction, lower_order_final, no Karras sigmas; diffusers itself is absent here: the solver pieces are checked through
    the first order == DDIM identity, and the inverse scheduler built on them is pinned by the reference's own class (tests/golden/dpm_inverse.npz).  The latent update is one
    etainv_lincomb3 launch; all coefficients are host float64 scalars."""
    Output = namedtuple("DPMSolverMultistepSchedulerOutput", ("prev_sample",))
    order = 2

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", solver_order=2,
                 prediction_type="epsilon", algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
                 timestep_spacing="leading", steps_offset=0, **extra):
        if beta_schedule != "scaled_linear" or prediction_type != "epsilon" or algorithm_type != "dpmsolver++" or solver_type != "midpoint" \
                or solver_order != 2 or extra.get("use_karras_sigmas") or extra.get("thresholding"):
            raise NotImplementedError("only the SD1.x DPM-Solver++(2M) configuration is built (scaled_linear, epsilon, dpmsolver++, midpoint, order 2)")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                              solver_order=solver_order, prediction_type=prediction_type, algorithm_type=algorithm_type, solver_type=solver_type,
                              lower_order_final=lower_order_final, timestep_spacing=timestep_spacing, steps_offset=steps_offset, **extra)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        ac = self.alphas_cumprod.numpy().astype(np.float64)
        self.alpha_t, self.sigma_t = np.sqrt(ac), np.sqrt(1.0 - ac)
        self.lambda_t = np.log(self.alpha_t) - np.log(self.sigma_t)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self.model_outputs, self.lower_order_nums = [None] * solver_order, 0
        self.final_timestep = 0                               # where the step after the last loop timestep lands

    @classmethod
    def from_config(cls, config):
        return cls(**{k: v for k, v in dict(config).items() if k not in ("clip_sample", "set_alpha_to_one")})

    def _grid(self, n):
        N = self.config.num_train_timesteps
        if self.config.timestep_spacing == "linspace":
            return np.linspace(0, N - 1, n + 1).round()
        if self.config.timestep_spacing == "leading":
            return (np.arange(0, n + 1) * (N // (n + 1))).round() + self.config.steps_offset
        raise NotImplementedError(self.config.timestep_spacing)

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(self._grid(num_inference_steps)[::-1][:-1].copy().astype(np.int64))
        self.model_outputs, self.lower_order_nums = [None] * self.config.solver_order, 0

    # ---- the three pieces of diffusers' step(), on device tensors through the C ABI
    def convert_model_output(self, model_output, timestep, sample):
        t = int(timestep)
        return _lincomb(sample, 1.0 / self.alpha_t[t], model_output, -self.sigma_t[t] / self.alpha_t[t])           # x0 prediction

    def dpm_solver_first_order_update(self, m0, timestep, prev_timestep, sample):
        s, t = int(timestep), int(prev_timestep)
        h = self.lambda_t[t] - self.lambda_t[s]
        return _lincomb(sample, self.sigma_t[t] / self.sigma_t[s], m0, -self.alpha_t[t] * (np.exp(-h) - 1.0))

    def multistep_dpm_solver_second_order_update(self, model_output_list, timestep_list, prev_timestep, sample):
        t, s0, s1 = int(prev_timestep), int(timestep_list[-1]), int(timestep_list[-2])
        m0, m1 = model_output_list[-1], model_output_list[-2]
        h, h0 = self.lambda_t[t] - self.lambda_t[s0], self.lambda_t[s0] - self.lambda_t[s1]
        c = self.alpha_t[t] * (np.exp(-h) - 1.0)
        k = 0.5 * c * (h / h0)                                 # 0.5 c D1, D1 = (m0 - m1) / r0, r0 = h0 / h
        return _lincomb(sample, self.sigma_t[t] / self.sigma_t[s0], m0, -c - k, m1, k)

    def _advance(self, x0, step_index, timestep, sample):
        n = len(self.timesteps)
        prev_timestep = self.final_timestep if step_index == n - 1 else int(self.timesteps[step_index + 1])
        lower_order_final = step_index == n - 1 and self.config.lower_order_final and n < 15
        self.model_outputs = self.model_outputs[1:] + [x0]
        if self.lower_order_nums < 1 or lower_order_final:
            out = self.dpm_solver_first_order_update(x0, timestep, prev_timestep, sample)
        else:
            out = self.multistep_dpm_solver_second_order_update(self.model_outputs, [int(self.timesteps[step_index - 1]), timestep], prev_timestep, sample)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        return out

    def step(self, model_output, timestep, sample, **unused):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        idx = (self.timesteps == int(timestep)).nonzero()
        step_index = len(self.timesteps) - 1 if len(idx) == 0 else int(idx[0])
        x0 = self.convert_model_output(model_output, int(timestep), sample)
        return DPMSolverMultistepScheduler.Output(self._advance(x0, step_index, int(timestep), sample))


def _lincomb(x, a, y, b, z=None, c=0.0):
    x, y = x.contiguous(), y.contiguous()
    z = None if z is None else z.contiguous()
    out = torch.empty_like(x)
    _capi.check(_capi.load().etainv_lincomb3(_capi.ptr(x), float(a), _capi.ptr(y), float(b), _capi.ptr(z), float(c), _capi.ptr(out), x.numel(),
                                            _capi.dtype_code(x.dtype), _capi.stream_ptr()))
    return out
