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
        if guidance_scale is None:
            return self.unet(latent, t, encoder_hidden_states=context, **kwargs)["sample"]
        if latent.shape[0] * 2 == context.shape[0]:
            latent = torch.cat([latent] * 2)
        else:
            assert latent.shape[0] == context.shape[0]
        n = latent.shape[0] // 2
        if isinstance(guidance_scale, (int, float)) and guidance_scale == 0:
            return self.unet(latent[:n], t, encoder_hidden_states=context[:n], **kwargs)["sample"]
        if isinstance(guidance_scale, (int, float)) and guidance_scale == 1:
            return self.unet(latent[n:], t, encoder_hidden_states=context[n:], **kwargs)["sample"]
        return self._cfg(self.unet(latent, t, encoder_hidden_states=context, **kwargs)["sample"], guidance_scale)

    def step_forward(self, noise_pred, t, latent, *args, **kwargs) -> Any:
        return self.scheduler_fwd.step(noise_pred, t, latent, *args, **kwargs)

    def step_backward(self, noise_pred, t, latent, *args, **kwargs) -> Any:
        return self.scheduler_bwd.step(noise_pred, t, latent, *args, **kwargs)

    def predict_step_forward(self, latent, t, context, guidance_scale_fwd=None):
        guidance_scale_fwd = guidance_scale_fwd or self.guidance_scale_fwd
        latent = self.controller.begin_step(latent=latent)
        noise_pred = self.predict_noise(latent, t, context, guidance_scale_fwd, is_fwd=True)
        new_latent = self.step_forward(noise_pred, t, latent).prev_sample
        new_latent = self.controller.end_step(latent=new_latent, noise_pred=noise_pred, t=t)
        return new_latent, noise_pred

    def predict_step_backward(self, latent, t, context, guidance_scale_bwd=None):
        guidance_scale_bwd = guidance_scale_bwd or self.guidance_scale_bwd
        latent = self.controller.begin_step(latent=latent, t=t)
        noise_pred = self.predict_noise(latent, t, context, guidance_scale_bwd)
        new_latent = self.step_backward(noise_pred, t, latent).prev_sample
        new_latent = self.controller.end_step(latent=new_latent, noise_pred=noise_pred, t=t)
        return new_latent, noise_pred

    def get_timesteps_forward(self) -> torch.Tensor:
        return self.scheduler_fwd.timesteps

    def get_timesteps_backward(self) -> torch.Tensor:
        return self.scheduler_bwd.timesteps

    # ------------------------------------------------------------------ loops
    def diffusion_forward(self, latent, context, guidance_scale_fwd=None) -> Dict[str, Any]:
        guidance_scale_fwd = guidance_scale_fwd or self.guidance_scale_fwd
        latents, noise_preds = [latent], []
        latent = latent.clone().detach()
        for t in self.pbar(self.get_timesteps_forward(), desc="forward"):
            latent, noise_pred = self.predict_step_forward(latent, t, context, guidance_scale_fwd)
            noise_preds.append(noise_pred)
            latents.append(latent)
        return {"latents": latents, "noise_preds": noise_preds, "zT_inv": latents[-1]}

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
        n, b = len(contexts), contexts[0].shape[0]
        assert b == 2, "Cfg should have batch dimension 2"
        x = torch.stack(contexts, 1)
        return x.reshape(b * n, *x.shape[2:])

    def cat_latent(self, latents: List[torch.Tensor]) -> torch.Tensor:
        return torch.cat(latents)

    def sample(self, inv_result, prompt=None, context=None):
        if inv_result is None:
            return None
        latent = inv_result["latents"][-1]
        context = context if context is not None else self.create_context(prompt)
        if isinstance(context, list):
            n = len(context)
            context = self.cat_context(context)
            latent = self.cat_latent([latent] * n)
        z0 = self.diffusion_backward(latent, context, inv_result)
        if z0 is None:
            return None
        return {"image": self.decode(z0), "latent": z0}

    def invert_sample(self, image, prompt: str):
        context = self.create_context(prompt)
        return self.sample(self.invert(image, context=context), context=context)
