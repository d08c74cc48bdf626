"""Eta Inversion on the native engine.  Plugin surface of the reference's modules/inversion/eta_inversion.py:61-404
(same constructor arguments and defaults, `invert` / `sample` / `predict_step_backward` / `get_eta_variance_noise` /
`sample_variance_noise` / `get_mask`), without its import-time side effect (:19-20), `eval` (:54-56) and the
`EtaTensor` hack (:23-33).  `invert` + `sample` with the built-in editors run the batched device loops of
`etainv.pipeline.EtaLoop`; the per-step methods call the same C-ABI kernels one step at a time."""
from typing import Any, Dict, Optional

import numpy as np
import torch

from etainv import _capi
from etainv.pipeline import EtaLoop, PtpTables, eta_table
from ..editing.controller import ControllerEmpty
from .diffusion_inversion import DiffusionInversion


class EtaInversion(DiffusionInversion):
    def __init__(self, model, scheduler: Optional[str] = None, num_inference_steps: Optional[int] = None,
                 guidance_scale_bwd: Optional[float] = None, guidance_scale_fwd: Optional[float] = None, verbose: bool = False,
                 eta=(0.0, 0.4), noise_sample_count: int = 10, seed: int = 0, eta_start: Optional[float] = None,
                 eta_end: Optional[float] = None, use_mask=True, mask_mode_cfg=None) -> None:
        if use_mask:
            dft = dict(attn_from_where=["up", "down"], attn_res=16, mask_dirinv=None, mask_eta="fwd_mean", pow=None,
                       target_dirinv=None, thres=0.2)
            mask_mode_cfg = {**dft, **(mask_mode_cfg or {})}
            if mask_mode_cfg["mask_eta"] not in ("fwd_mean", "fwd", "gt", "bwd_source", "bwd_target", "bwd_source_target") \
                    or mask_mode_cfg["mask_dirinv"] not in (None, mask_mode_cfg["mask_eta"]):
                raise NotImplementedError("eta-mask sources built: fwd_mean (default), fwd, gt, bwd_source, bwd_target, bwd_source_target, each "
                                          "with thres / pow; mask_dirinv must be None or the same source as mask_eta")
        else:
            mask_mode_cfg = None
        self.mask_mode_cfg = mask_mode_cfg
        g_fwd_pair = None
        if isinstance(guidance_scale_fwd, (tuple, list)):                  # per-timestep table (reference :108-110): handled by the native loop
            assert len(guidance_scale_fwd) == 2
            g_fwd_pair, guidance_scale_fwd = tuple(guidance_scale_fwd), None
        super().__init__(model, scheduler, num_inference_steps, guidance_scale_bwd, guidance_scale_fwd, verbose)
        self._g_fwd_pair = g_fwd_pair
        if eta_start is not None:
            assert eta_end is not None
            eta = (eta_start, eta_end)
        self.etas = eta_table(eta)
        self.attn_maps_forward = {}
        self._step_mask = None
        self.noise_sample_count = noise_sample_count
        self.seed = seed if seed >= 0 else None
        self.L = model.engine.L
        self._loop = EtaLoop(model.engine, S=self.num_inference_steps, guidance_scale_bwd=self.guidance_scale_bwd,
                             guidance_scale_fwd=self._g_fwd_pair or self.guidance_scale_fwd, eta=eta, noise_sample_count=noise_sample_count,
                             use_mask=use_mask, mask_thres=(mask_mode_cfg or {}).get("thres", 0.2),
                             mask_eta=(mask_mode_cfg or {}).get("mask_eta", "fwd_mean"), mask_pow=(mask_mode_cfg or {}).get("pow"),
                             target_dirinv=(mask_mode_cfg or {}).get("target_dirinv"), mask_dirinv=(mask_mode_cfg or {}).get("mask_dirinv"))

    # ------------------------------------------------------------------ noise / mask
    def sample_variance_noise(self, n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """(n,1,4,L,L) candidates.  Drawn on the CPU generator (reproducible across devices; the reference draws on the
        model device, eta_inversion.py:156) and moved to the device."""
        return torch.randn((n, 1, 4, self.L, self.L), generator=generator).to(self.model.device)

    def get_mask(self, key, mask, t, edit_word_idx):
        if self.mask_mode_cfg is None or self.mask_mode_cfg[key] is None:
            return None
        mode = self.mask_mode_cfg[key]                                 # reference eta_inversion.py:159-205
        if mode.startswith("bwd"):
            raise NotImplementedError("bwd_* eta masks are built for the fused loop (etainv + ptp editor), not for the per-step API")
        if mode == "gt":
            m = mask
        elif mode == "fwd":
            m = self.attn_maps_forward[int(t)][edit_word_idx[0]]
        else:
            m = self.attn_maps_forward["mean"][edit_word_idx[0]]
        if self.mask_mode_cfg["thres"] is not None:
            m = (m > self.mask_mode_cfg["thres"]).to(m.dtype)
        if self.mask_mode_cfg["pow"] is not None:
            m = torch.pow(m, self.mask_mode_cfg["pow"])
        return m

    # ------------------------------------------------------------------ inversion
    def _word_tokens(self, prompt):
        words = prompt.split(" ")
        return torch.tensor([[words.index(w) + 1 for w in words]], dtype=torch.int32, device=self.model.device)   # ptp_editor.py:72

    def invert(self, image, prompt=None, context=None, guidance_scale_fwd=None, inv_cfg: Optional[Dict[str, Any]] = None):
        if self.mask_mode_cfg is not None:
            if inv_cfg["edit_word_idx"][0] is None or inv_cfg["edit_word_idx"][1] is None:
                return None                                            # eta_inversion.py:385-386
        context = context if context is not None else self.create_context(prompt)
        z0 = self.encode(image).float().contiguous()
        tokens = self._word_tokens(prompt) if self.mask_mode_cfg is not None else None
        res = self._loop.invert(z0, context[None].float(), tokens)
        lat = res["latents"]                                           # (S+1, 1, 4, L, L)
        out = {"inv_cfg": inv_cfg, "latents": [lat[j] for j in range(lat.shape[0])], "noise_preds": None, "zT_inv": lat[-1],
               "context": context, "_native": res}
        self.attn_maps_forward = {}
        if res["maps_mean"] is not None:
            self.attn_maps_forward["mean"] = [res["maps_mean"][0, w][None] for w in range(res["maps_mean"].shape[1])]
        if res.get("maps_steps") is not None:                              # keyed by timestep like the reference (:44-49)
            for j, tt in enumerate(self._loop.t_fwd):
                self.attn_maps_forward[int(tt)] = [res["maps_steps"][j, 0, w][None] for w in range(res["maps_steps"].shape[2])]
        return out

    # ------------------------------------------------------------------ backward
    def _tables_from_controller(self):
        """(PtpTables | None, masactrl | None, fast_path_ok)"""
        from ..editing.ptp_editor import PromptToPromptController
        from ..editing.masactrl_editor import MasactrlController
        c = self.controller
        if isinstance(c, ControllerEmpty):
            return None, None, True
        if isinstance(c, PromptToPromptController):
            t = c.controller.tables()
            st = lambda a: None if a is None else a[None]
            ptp = PtpTables(st(t["mapper"]), st(t["alphas"]), t["cross_alpha"][:, None], c.controller.self_replace_steps,
                            self.num_inference_steps, equalizer=st(t["equalizer"]), blend_alpha=st(t["blend_alpha"]),
                            replace_mat=st(t["replace_mat"]), device=self.model.device)
            return ptp, None, True
        if isinstance(c, MasactrlController):
            return None, (c.step, c.layer), True
        return None, None, False

    def diffusion_backward(self, latent, context, inv_result):
        S, L = self.num_inference_steps, self.L
        inv_cfg = inv_result.get("inv_cfg") or {}
        edit_word_idx = inv_cfg.get("edit_word_idx", None)
        ptp, masa, fast = self._tables_from_controller()
        generator = torch.Generator().manual_seed(self.seed) if self.seed is not None else None
        if fast and latent.shape[0] == 2 and "_native" in inv_result:
            noise = torch.stack([self.sample_variance_noise(self.noise_sample_count, generator) for _ in range(S)])
            noise = noise.reshape(S, self.noise_sample_count, 4, L, L).contiguous()
            ctx = context.reshape(2, 2, *context.shape[1:])            # [half][role]
            ctx_src, ctx_tgt = ctx[:, 0][None], ctx[:, 1][None]
            ew = torch.tensor([edit_word_idx[0]]) if self.mask_mode_cfg is not None else None
            gt = None
            if self.mask_mode_cfg is not None and self.mask_mode_cfg["mask_eta"] == "gt":
                gt = inv_cfg["mask"]                                       # bilinear to the latent grid (eta_inversion.py:286-287)
                gt = torch.nn.functional.interpolate(gt.float().reshape(1, 1, *gt.shape[-2:]), (L, L), mode="bilinear")[0]
            ew_t = torch.tensor([edit_word_idx[1]]) if self.mask_mode_cfg is not None else None
            return self._loop.sample(inv_result["_native"], ctx_src, ctx_tgt, noise, edit_word=ew, ptp=ptp, masactrl=masa, gt_mask=gt,
                                     edit_word_tgt=ew_t)
        # generic path: user-defined controllers keep their per-step callbacks
        mask = inv_cfg.get("mask", None)
        if mask is not None:                                               # eta_inversion.py:286-287
            mask = torch.nn.functional.interpolate(mask.float().reshape(1, 1, *mask.shape[-2:]), (L, L), mode="bilinear")[0].to(self.model.device)
        for i, t in enumerate(self.pbar(self.scheduler_bwd.timesteps, desc="backward")):
            latent, _ = self.predict_step_backward(latent, t, context, source_latent_prev=inv_result["latents"][-(i + 2)],
                                                   generator=generator, mask=mask, edit_word_idx=edit_word_idx)
        return latent

    def predict_step_backward(self, latent, t, context, guidance_scale_bwd=None, source_latent_prev=None, generator=None, mask=None,
                              edit_word_idx=None):
        guidance_scale_bwd = guidance_scale_bwd or self.guidance_scale_bwd
        latent = self.controller.begin_step(latent=latent, t=t)
        assert latent.shape[0] == 2 and context.shape[0] == 4, "one (source, target) pair"
        eps_all = self.unet(torch.cat([latent] * 2), t, encoder_hidden_states=context)["sample"].float().contiguous()
        self._step_mask = mask
        res = self.get_eta_variance_noise(source_latent_prev, latent, t, eps_all, generator, _fused=True, edit_word_idx=edit_word_idx)
        new_latent = self.controller.end_step(latent=res["latent"], noise_pred=res["noise_pred"], t=t)
        return new_latent, res["noise_pred"]

    def get_eta_variance_noise(self, latent_prev, latent, t, noise_pred, generator=None, _fused=False, edit_word_idx=None):
        """Fused CFG + best-of-n + masked eta step (etainv_eta_backward_step).  `noise_pred` = raw UNet output rows
        [u_s,u_t,c_s,c_t]; returns eta, the chosen variance noise, its index and the updated latents."""
        t = int(t)
        S, L = self.num_inference_steps, self.L
        cand = self.sample_variance_noise(self.noise_sample_count, generator).reshape(self.noise_sample_count, 4, L, L).float().contiguous()
        sch = self.scheduler_bwd
        p = t - sch.config.num_train_timesteps // S
        a_t, a_p, var = sch._alpha(t), sch._alpha(p), sch._get_variance(t, p)
        use_mask = self.mask_mode_cfg is not None
        mask_map = self.get_mask("mask_eta", self._step_mask, t, edit_word_idx).float().reshape(1, L, L).contiguous() if use_mask else None
        x = latent.float().contiguous()
        out_x, out_eps = torch.empty_like(x), torch.empty_like(x)
        best = torch.zeros(1, dtype=torch.int32, device=x.device)
        losses = torch.zeros(1, self.noise_sample_count, dtype=torch.float32, device=x.device)
        scratch = torch.empty(16 * 64, dtype=torch.float32, device=x.device)
        tdir = (self.mask_mode_cfg or {}).get("target_dirinv")
        dmap = None
        if tdir is not None and self.mask_mode_cfg["mask_dirinv"] is not None:
            dmap = (1.0 - self.get_mask("mask_dirinv", self._step_mask, t, edit_word_idx).float().reshape(1, L, L)).contiguous()
        _capi.check(_capi.load().etainv_eta_backward_step_ex(
            _capi.ptr(x), _capi.ptr(noise_pred), float(self.guidance_scale_bwd), _capi.ptr(latent_prev.float().contiguous()),
            _capi.ptr(cand), self.noise_sample_count, float(self.etas[t]), _capi.ptr(mask_map),
            0.0, 2 if use_mask else 0, a_t, a_p, var, 1, 4, L * L, _capi.ptr(out_x),   # 2: get_mask already applied thres / pow
            _capi.ptr(out_eps), _capi.ptr(best), _capi.ptr(losses), _capi.ptr(scratch), _capi.F32, float(tdir or 0.0), _capi.ptr(dmap),
            _capi.stream_ptr()))
        return {"eta": float(self.etas[t]), "variance_noise_candidates": cand, "best_idx": best, "losses": losses, "latent": out_x,
                "noise_pred": out_eps, "latent_prev": latent_prev}
