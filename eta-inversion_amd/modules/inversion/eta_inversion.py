"""Eta Inversion on the native engine.  Plugin surface of the reference's modules/inversion/eta_inversion.py:61-404
(same constructor arguments and defaults, `invert` / `sample` / `predict_step_backward` / `get_eta_variance_noise` /
`sample_variance_noise` / `get_mask`), without its import-time side effect (:19-20), `eval` (:54-56) and the
`EtaTensor` hack (:23-33).  `invert` + `sample` with the built-in editors run the batched device loops of
`etainv.pipeline.EtaLoop`; the per-step methods call the same C-ABI kernels one step at a time."""
from typing import Any, Dict, Optional

import numpy as np
import torch

from etainv import _capi
from etainv.pipeline import EtaLoop, PtpTables, attn_layer_selection, eta_table
from ..editing.controller import ControllerEmpty
from .diffusion_inversion import DiffusionInversion


class EtaInversion(DiffusionInversion):
    def __init__(self, model, scheduler: Optional[str] = None, num_inference_steps: Optional[int] = None,
                 guidance_scale_bwd: Optional[float] = None, guidance_scale_fwd: Optional[float] = None, verbose: bool = False,
                 eta=(0.0, 0.4), noise_sample_count: int = 10, seed: int = 0, eta_start: Optional[float] = None,
                 eta_end: Optional[float] = None, use_mask=True, mask_mode_cfg=None) -> None:
        if use_mask:
            # attn_res: the reference's 16 is 64 / 4 (eta_inversion.py:90); generalised to L / 4 for other latent sizes
            dft = dict(attn_from_where=["up", "down"], attn_res=model.engine.L // 4, mask_dirinv=None, mask_eta="fwd_mean", pow=None,
                       target_dirinv=None, thres=0.2)
            mask_mode_cfg = {**dft, **(mask_mode_cfg or {})}
            srcs = ("fwd_mean", "fwd", "gt", "bwd_source", "bwd_target", "bwd_source_target")
            if mask_mode_cfg["mask_eta"] not in srcs + (None,) or mask_mode_cfg["mask_dirinv"] not in srcs + (None,):
                raise ValueError(f"mask_eta / mask_dirinv must be one of {srcs} or None (reference eta_inversion.py:164-187)")
            # the engine's store keeps the cross layers of ONE resolution per pass (etainv_maps_configure): L/4 (default), L/2 or L/8 for the forward-pass
            # sources; the backward pass always keeps L/4 (LocalBlend), so bwd_* sources take `attn_from_where` subsets at L/4 only
            attn_layer_selection(model.engine.L, mask_mode_cfg["attn_res"], mask_mode_cfg["attn_from_where"])   # raises on what cannot be served
        else:
            mask_mode_cfg = None
        self.mask_mode_cfg = mask_mode_cfg
        g_fwd_pair = None
        if isinstance(guidance_scale_fwd, (tuple, list)):                  # per-timestep table (reference :108-110): handled by the native loop
            assert len(guidance_scale_fwd) == 2
            g_fwd_pair, guidance_scale_fwd = tuple(guidance_scale_fwd), None
        name = scheduler if isinstance(scheduler, str) or scheduler is None else scheduler.get("type")
        if name not in (None, "ddim"):                                     # the backward step passes eta / variance_noise (reference :245): DDIM only
            raise NotImplementedError(f"etainv / dirinv step the backward pass with DDIM(eta); scheduler '{name}' has no eta (use diffinv for 'dpm')")
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
                             use_mask=use_mask and mask_mode_cfg["mask_eta"] is not None, mask_thres=(mask_mode_cfg or {}).get("thres", 0.2),
                             mask_eta=(mask_mode_cfg or {}).get("mask_eta", "fwd_mean"), mask_pow=(mask_mode_cfg or {}).get("pow"),
                             target_dirinv=(mask_mode_cfg or {}).get("target_dirinv"), mask_dirinv=(mask_mode_cfg or {}).get("mask_dirinv"),
                             attn_res=(mask_mode_cfg or {}).get("attn_res"), attn_from_where=(mask_mode_cfg or {}).get("attn_from_where", ("up", "down")))

    # ------------------------------------------------------------------ noise / mask
    def sample_variance_noise(self, n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """(n,1,4,L,L) candidates.  Drawn on the CPU generator (reproducible across devices; the reference draws on the
        model device, eta_inversion.py:156) and moved to the device."""
        return torch.randn((n, 1, 4, self.L, self.L), generator=generator).to(self.model.device)

    def get_mask(self, key, mask, t, edit_word_idx):
        """reference eta_inversion.py:159-205: the raw map of the configured source, then thres / pow.  The bwd_* sources read the
        backward-pass store of the active prompt-to-prompt controller (its running average over the steps done, this one included)."""
        if self.mask_mode_cfg is None or self.mask_mode_cfg[key] is None:
            return None
        mode = self.mask_mode_cfg[key]
        if mode == "gt":
            m = mask
        elif mode == "fwd":
            m = self.attn_maps_forward[int(t)][edit_word_idx[0]]
        elif mode == "fwd_mean":
            m = self.attn_maps_forward["mean"][edit_word_idx[0]]
        else:
            if not hasattr(self.controller, "get_attention_map"):
                raise RuntimeError(f"mask source '{mode}' needs a prompt-to-prompt controller (its attention store), got {type(self.controller).__name__}")
            cfg = self.mask_mode_cfg
            amap = lambda word, prompt: self.controller.get_attention_map(mask_idx=word, res=cfg["attn_res"], from_where=cfg["attn_from_where"],
                                                                          prompt_idx=prompt, num_prompts=2, resize=self.L)
            if mode == "bwd_source":
                m = amap(edit_word_idx[0], 0)
            elif mode == "bwd_target":
                m = amap(edit_word_idx[1], 1)
            else:
                m = torch.maximum(amap(edit_word_idx[0], 0), amap(edit_word_idx[1], 1))
        if self.mask_mode_cfg["thres"] is not None:
            m = (m > self.mask_mode_cfg["thres"]).to(m.dtype)
        if self.mask_mode_cfg["pow"] is not None:
            m = torch.pow(m, self.mask_mode_cfg["pow"])
        return m

    # ------------------------------------------------------------------ inversion
    def _word_tokens(self, prompt):
        words = prompt.split(" ")
        if len(words) + 1 > 76:                                         # token index word + 1 must stay inside the 77-token context
            raise IndexError(f"prompt has {len(words)} whitespace words: word maps index the 77-token context (ptp.py:296)")
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
        if latent.shape[0] != 2:
            raise NotImplementedError("etainv / dirinv replay the source row next to the target row: the backward pass needs the [source, target] "
                                      "pair (no_source_backward editors only make sense with the plain inverters, e.g. diffinv)")
        S, L = self.num_inference_steps, self.L
        inv_cfg = inv_result.get("inv_cfg") or {}
        edit_word_idx = inv_cfg.get("edit_word_idx", None)
        ptp, masa, fast = self._tables_from_controller()
        generator = torch.Generator().manual_seed(self.seed) if self.seed is not None else None
        # force_per_step (attribute, default off): run the reference-style callback-per-step path also for the built-in controllers
        if fast and latent.shape[0] == 2 and "_native" in inv_result and not getattr(self, "force_per_step", False):
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
        """One backward step with the reference's call order (eta_inversion.py:207-273): controller.begin_step -> UNet (+ the controller's
        attention control) -> get_mask -> fused CFG + best-of-n + masked eta update + source replay -> controller.end_step."""
        guidance_scale_bwd = guidance_scale_bwd or self.guidance_scale_bwd
        latent = self.controller.begin_step(latent=latent, t=t)
        assert latent.shape[0] == 2 and context.shape[0] == 4, "one (source, target) pair"
        eps_all = self.unet(torch.cat([latent] * 2), t, encoder_hidden_states=context)["sample"].float().contiguous()
        res = self._fused_eta_step(source_latent_prev, latent, t, eps_all, generator, mask, edit_word_idx, guidance_scale_bwd)
        new_latent = self.controller.end_step(latent=res["latent"], noise_pred=res["noise_pred"], t=t)
        return new_latent, res["noise_pred"]

    def _fused_eta_step(self, latent_prev, latent, t, eps_all, generator, mask, edit_word_idx, guidance_scale_bwd=None):
        """etainv_eta_backward_step_ex on one pair: `eps_all` = raw UNet output rows [u_s,u_t,c_s,c_t], `latent` = [source, target]."""
        t = int(t)
        S, L = self.num_inference_steps, self.L
        cand = self.sample_variance_noise(self.noise_sample_count, generator).reshape(self.noise_sample_count, 4, L, L).float().contiguous()
        sch = self.scheduler_bwd
        p = t - sch.config.num_train_timesteps // S
        a_t, a_p, var = sch._alpha(t), sch._alpha(p), sch._get_variance(t, p)
        m_eta = self.get_mask("mask_eta", mask, t, edit_word_idx) if self.mask_mode_cfg is not None else None
        use_mask = self.mask_mode_cfg is not None
        if use_mask and m_eta is None:                                     # mask_eta None inside a mask_mode_cfg: eta everywhere (:239-240)
            m_eta = torch.ones(1, L, L, dtype=torch.float32, device=latent.device)
        mask_map = m_eta.float().reshape(1, L, L).contiguous() if use_mask else None
        x = latent.float().contiguous()
        out_x, out_eps = torch.empty_like(x), torch.empty_like(x)
        best = torch.zeros(1, dtype=torch.int32, device=x.device)
        losses = torch.zeros(1, self.noise_sample_count, dtype=torch.float32, device=x.device)
        scratch = torch.empty(16 * 64, dtype=torch.float32, device=x.device)
        tdir = (self.mask_mode_cfg or {}).get("target_dirinv")
        dmap = None
        if tdir is not None and self.mask_mode_cfg["mask_dirinv"] is not None:
            dmap = (1.0 - self.get_mask("mask_dirinv", mask, t, edit_word_idx).float().reshape(1, L, L)).contiguous()
        _capi.check(_capi.load().etainv_eta_backward_step_ex(
            _capi.ptr(x), _capi.ptr(eps_all), float(guidance_scale_bwd or self.guidance_scale_bwd), _capi.ptr(latent_prev.float().contiguous()),
            _capi.ptr(cand), self.noise_sample_count, float(self.etas[t]), _capi.ptr(mask_map),
            0.0, 2 if use_mask else 0, a_t, a_p, var, 1, 4, L * L, _capi.ptr(out_x),   # 2: get_mask already applied thres / pow
            _capi.ptr(out_eps), _capi.ptr(best), _capi.ptr(losses), _capi.ptr(scratch), _capi.F32, float(tdir or 0.0), _capi.ptr(dmap),
            _capi.stream_ptr()))
        return {"eta": float(self.etas[t]), "variance_noise_candidates": cand, "best_idx": best, "losses": losses, "latent": out_x,
                "noise_pred": out_eps, "latent_prev": latent_prev}

    def compute_optimal_variance_noise(self, latent_prev, latent, t, eta, noise_pred):
        """(latent_prev - DDIM-eta mean) / (eta sqrt(var)) -- reference eta_inversion.py:296-317 (eta == 0 gives inf / NaN like there)"""
        mean = self.step_backward(noise_pred, t, latent, eta=eta, variance_noise=torch.zeros_like(noise_pred)).prev_sample
        sch = self.scheduler_bwd
        var = sch._get_variance(int(t), int(t) - sch.config.num_train_timesteps // self.num_inference_steps)
        return (latent_prev - mean) / torch.tensor(eta * var ** 0.5, dtype=latent_prev.dtype, device=latent_prev.device)

    def get_eta_variance_noise(self, latent_prev, latent, t, noise_pred, generator=None):
        """Reference signature and result keys (eta_inversion.py:330-375): `latent` = the source row (1,4,L,L), `noise_pred` = its GUIDED
        noise.  Best-of-n selection runs in the fused kernel (fed with u = c = noise_pred so that its CFG combine is the identity); the
        chosen candidate is gathered on the device (no host sync)."""
        tt = int(t)
        S, L = self.num_inference_steps, self.L
        cand5 = self.sample_variance_noise(self.noise_sample_count, generator)                     # (n,1,4,L,L)
        cand = cand5.reshape(self.noise_sample_count, 4, L, L).float().contiguous()
        sch = self.scheduler_bwd
        p = tt - sch.config.num_train_timesteps // S
        a_t, a_p, var = sch._alpha(tt), sch._alpha(p), sch._get_variance(tt, p)
        x = torch.cat([latent, latent]).float().contiguous()
        eps4 = torch.cat([noise_pred] * 4).float().contiguous()
        out_x = torch.empty_like(x)
        best = torch.zeros(1, dtype=torch.int32, device=x.device)
        losses = torch.zeros(1, self.noise_sample_count, dtype=torch.float32, device=x.device)
        scratch = torch.empty(16 * 64, dtype=torch.float32, device=x.device)
        eta = float(self.etas[tt])
        _capi.check(_capi.load().etainv_eta_backward_step(
            _capi.ptr(x), _capi.ptr(eps4), 1.0, _capi.ptr(latent_prev.float().contiguous()), _capi.ptr(cand), self.noise_sample_count, eta,
            None, 0.0, 0, a_t, a_p, var, 1, 4, L * L, _capi.ptr(out_x), None, _capi.ptr(best), _capi.ptr(losses), _capi.ptr(scratch), _capi.F32,
            _capi.stream_ptr()))
        variance_noise = cand5.index_select(0, best.long()).reshape(1, 4, L, L)
        latent_prev_rec = self.step_backward(noise_pred, t, latent, eta=eta, variance_noise=variance_noise).prev_sample
        return {"eta": eta, "variance_noise": variance_noise, "delta": latent_prev - latent_prev_rec, "latent_prev": latent_prev,
                "latent_prev_rec": latent_prev_rec, "loss": losses[0].index_select(0, best.long())[0], "best_idx": best, "losses": losses}
