"""Base inverter with the reference's plugin surface (modules/inversion/diffusion_inversion.py:12-542), backed by the
native engine: `predict_noise` -> etainv_unet_forward (+ etainv_cfg_combine), `step_forward` -> etainv_ddim_step,
`step_backward` -> etainv_ddim_eta_step.  Method names, arguments and result dictionaries follow the reference."""
import contextlib
from typing import Any, Dict, Iterator, List, Optional

import torch

from etainv import _capi
from ..editing.controller import ControllerBase, ControllerEmpty
from ..inverse_schedulers import DDIMInverseScheduler, DPMSolverMultistepInverseScheduler
from ..schedulers import DDIMScheduler, DPMSolverMultistepScheduler


class DiffusionInversion:
    def __init__(self, model, scheduler: Optional[str] = None, num_inference_steps: Optional[int] = None,
                 guidance_scale_bwd: Optional[float] = None, guidance_scale_fwd: Optional[float] = None, verbose: bool = False) -> None:
        scheduler = scheduler or "ddim"
        self.num_inference_steps = num_inference_steps or 50
        self.guidance_scale_bwd = guidance_scale_bwd if guidance_scale_bwd is not None else 7.5
        self.guidance_scale_fwd = guidance_scale_fwd if guidance_scale_fwd is not None else 1
        self.model, self.unet, self.device, self.verbose = model, model.unet, model.device, verbose
        self.controller = None
        model.scheduler, self.scheduler_bwd, self.scheduler_fwd = self.create_schedulers(model, scheduler, self.num_inference_steps)
        self.bwd_t_to_i = {t.item(): i for i, t in enumerate(self.scheduler_bwd.timesteps)}
        self.fwd_t_to_i = {t.item(): i for i, t in enumerate(self.scheduler_fwd.timesteps)}
        with self.use_controller(None):
            pass

    @contextlib.contextmanager
    def use_controller(self, controller: Optional[ControllerBase]) -> Iterator[None]:
        self.controller = controller if controller is not None else ControllerEmpty()
        self.controller.begin()
        yield
        self.controller.end()
        self.controller = ControllerEmpty()

    def pbar(self, it, **kwargs):
        if self.verbose:
            from tqdm import tqdm
            return tqdm(it, **kwargs)
        return it

    def create_schedulers(self, model, scheduler, num_inference_steps, scheduler_inv_kwargs=None):
        scheduler_inv_kwargs = dict(scheduler_inv_kwargs or {})
        if isinstance(scheduler, str):
            name, kw = scheduler, {}
        elif isinstance(scheduler, dict):
            kw = {**scheduler}
            name = kw.pop("type")
            if "inv_steps" in kw:
                scheduler_inv_kwargs["inv_steps"] = kw.pop("inv_steps")
        else:
            raise Exception(type(scheduler))
        if name == "dpm":                                                  # DPM-Solver++(2M) pair (reference :139-165)
            sched = DPMSolverMultistepScheduler.from_config({**model.scheduler.config, **kw})
            sched.set_timesteps(num_inference_steps)
            fwd = DPMSolverMultistepInverseScheduler.from_scheduler(sched, **scheduler_inv_kwargs)
            fwd.set_timesteps(num_inference_steps)
            assert fwd.timesteps[0] < fwd.timesteps[1], "wrong timestamp order, not increasing"
            return sched, sched, fwd
        if name != "ddim":
            raise NotImplementedError(f"scheduler '{name}' is not built in the MI355X engine (built: 'ddim', 'dpm'; 'ddpm' needs the stochastic "
                                      "DDPM inverse scheduler, outside the etainv hot path)")
        kw = {"clip_sample": False, "set_alpha_to_one": False, **kw}
        sched = DDIMScheduler.from_config({**model.scheduler.config, **kw})
        sched.set_timesteps(num_inference_steps)
        fwd = DDIMInverseScheduler.from_scheduler(sched, **scheduler_inv_kwargs)
        fwd.set_timesteps(num_inference_steps)
        assert fwd.timesteps[0] < fwd.timesteps[1], "wrong timestamp order, not increasing"
        return sched, sched, fwd

    @staticmethod
    def get_available_schedulers() -> List[str]:
        return ["ddim", "ddpm", "dpm"]

    # ------------------------------------------------------------------ VAE / text (third-party nets, outside the loop)
    def decode(self, latent: torch.Tensor) -> torch.Tensor:
        return self.model.vae.decode(1 / 0.18215 * latent)["sample"]

    def encode(self, image: torch.Tensor) -> torch.Tensor:
        return self.model.vae.encode(image.to(self.model.vae.dtype))["latent_dist"].mean * 0.18215

    def create_context(self, prompt: str, negative_prompt: Optional[str] = "") -> torch.Tensor:
        tok = self.model.tokenizer

        def embed(p):
            ids = tok([p], padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
            return self.model.text_encoder(ids.to(self.model.device))[0].float()
        cond = embed(prompt)
        if negative_prompt is None:
            return cond
        return torch.cat([embed(negative_prompt), cond])

    # ------------------------------------------------------------------ one step
    def _cfg(self, noise_pred, guidance_scale):
        u, c = noise_pred.chunk(2)
        u, c = u.contiguous(), c.contiguous()
        out = torch.empty_like(u)
        _capi.check(_capi.load().etainv_cfg_combine(_capi.ptr(u), _capi.ptr(c), float(guidance_scale), _capi.ptr(out), u.numel(),
                                                   _capi.dtype_code(u.dtype), _capi.stream_ptr()))
        return out

    def predict_noise(self, latent, t, context, guidance_scale, is_fwd: bool = False, **kwargs):
        """eps(latent, t, context) with classifier-free guidance (reference :263-286): guidance None = plain UNet call; a context twice as long as the
        latent batch = [uncond rows, cond rows] over the same latents; the scales 0 / 1 need one half only"""
        def eps(x, ctx):
            return self.unet(x, t, encoder_hidden_states=ctx, **kwargs)["sample"]

        if guidance_scale is None:
            return eps(latent, context)
        rows = context.shape[0]
        if rows == 2 * latent.shape[0]:
            latent = latent.repeat(2, *([1] * (latent.dim() - 1)))
        assert latent.shape[0] == rows
        half = rows // 2
        if isinstance(guidance_scale, (int, float)) and guidance_scale in (0, 1):
            keep = slice(half, rows) if guidance_scale == 1 else slice(0, half)
            return eps(latent[keep], context[keep])
        return self._cfg(eps(latent, context), guidance_scale)

    def step_forward(self, noise_pred, t, latent, *args, **kwargs) -> Any:
        return self.scheduler_fwd.step(noise_pred, t, latent, *args, **kwargs)

    def step_backward(self, noise_pred, t, latent, *args, **kwargs) -> Any:
        return self.scheduler_bwd.step(noise_pred, t, latent, *args, **kwargs)

    def _hooked_step(self, latent, t, context, scale, forward: bool):
        """one scheduler step between the controller's begin_step / end_step hooks (reference :328-341); the forward hook gets no timestep"""
        latent = self.controller.begin_step(latent=latent) if forward else self.controller.begin_step(latent=latent, t=t)
        eps = self.predict_noise(latent, t, context, scale, is_fwd=True) if forward else self.predict_noise(latent, t, context, scale)
        stepped = (self.step_forward if forward else self.step_backward)(eps, t, latent).prev_sample
        return self.controller.end_step(latent=stepped, noise_pred=eps, t=t), eps

    def predict_step_forward(self, latent, t, context, guidance_scale_fwd=None):
        return self._hooked_step(latent, t, context, guidance_scale_fwd or self.guidance_scale_fwd, True)

    def predict_step_backward(self, latent, t, context, guidance_scale_bwd=None):
        return self._hooked_step(latent, t, context, guidance_scale_bwd or self.guidance_scale_bwd, False)

    def get_timesteps_forward(self) -> torch.Tensor:
        return self.scheduler_fwd.timesteps

    def get_timesteps_backward(self) -> torch.Tensor:
        return self.scheduler_bwd.timesteps

    # ------------------------------------------------------------------ loops
    def diffusion_forward(self, latent, context, guidance_scale_fwd=None) -> Dict[str, Any]:
        """the inversion trajectory z_0 .. z_T with the noise predicted at each step (reference :401-418)"""
        scale = guidance_scale_fwd or self.guidance_scale_fwd
        trajectory, eps_seen = [latent], []
        z = latent.clone().detach()
        for t in self.pbar(self.get_timesteps_forward(), desc="forward"):
            z, eps = self.predict_step_forward(z, t, context, scale)
            trajectory.append(z)
            eps_seen.append(eps)
        return {"latents": trajectory, "noise_preds": eps_seen, "zT_inv": trajectory[-1]}

    def diffusion_backward(self, latent, context, inv_result):
        for t in self.pbar(self.get_timesteps_backward(), desc="backward"):
            latent, _ = self.predict_step_backward(latent, t, context)
        return latent

    def invert(self, image, prompt=None, context=None, guidance_scale_fwd=None, **kwargs) -> Dict[str, Any]:
        context = context if context is not None else self.create_context(prompt)
        latent = self.encode(image).float()
        res = self.diffusion_forward(latent, context, guidance_scale_fwd=guidance_scale_fwd)
        res["context"] = context
        return {**kwargs, **res}

    def cat_context(self, contexts: List[torch.Tensor]) -> torch.Tensor:
        """[(uncond, cond)] x n -> rows [uncond x n, cond x n] (reference :476-479)"""
        if any(c.shape[0] != 2 for c in contexts):
            raise AssertionError("Cfg should have batch dimension 2")
        return torch.stack(contexts, dim=1).flatten(0, 1)

    def cat_latent(self, latents: List[torch.Tensor]) -> torch.Tensor:
        return torch.cat(latents)

    def sample(self, inv_result, prompt=None, context=None):
        """backward pass from the inverted latent under `context` (a list = several prompts over copies of the latent; reference :507-528)"""
        if inv_result is None:
            return None
        if context is None:
            context = self.create_context(prompt)
        z_T = inv_result["latents"][-1]
        if isinstance(context, list):
            z_T = self.cat_latent([z_T] * len(context))
            context = self.cat_context(context)
        z0 = self.diffusion_backward(z_T, context, inv_result)
        return None if z0 is None else {"image": self.decode(z0), "latent": z0}

    def invert_sample(self, image, prompt: str):
        context = self.create_context(prompt)
        return self.sample(self.invert(image, context=context), context=context)
