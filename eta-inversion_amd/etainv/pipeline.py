"""Batched forward (inversion) / backward (eta-sampling) loops on the native engine.

Host logic only: timestep tables, per-step scalar coefficients and the order of C-ABI calls.  All tensors stay on
the device; nothing in the loops synchronises with the host (the best-of-n argmin runs in the eta-step kernel).

Replaces, for B independent (source, target) pairs at once (the reference handles exactly one, SURVEY E-6):
  DiffusionInversion.diffusion_forward / predict_step_forward   reference modules/inversion/diffusion_inversion.py:314-418
  EtaInversion.invert (per-step word maps + mean)               reference modules/inversion/eta_inversion.py:36-49,378-404
  EtaInversion.diffusion_backward / predict_step_backward       reference modules/inversion/eta_inversion.py:207-294
  DiffusionInversion.sample batch layout                        reference modules/inversion/diffusion_inversion.py:462-528
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi
from .engine import AttnControl
from .flops import exit_share

NUM_TRAIN = 1000


def alphas_cumprod() -> np.ndarray:
    """scaled_linear 0.00085..0.012, fp32 cumprod (reference modules/models/__init__.py:134)."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, NUM_TRAIN, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).numpy().astype(np.float64)


def eta_table(eta=(0.0, 0.4)) -> np.ndarray:
    """etas[1000] by raw timestep (reference modules/inversion/eta_inversion.py:52-58,121-139), without eval()."""
    if not isinstance(eta, (tuple, list)):
        eta = (eta, eta)
    ts = np.linspace(0, 1, NUM_TRAIN)
    if len(eta) == 3 or isinstance(eta[0], (tuple, list)):
        (x1, y1), (x2, y2) = eta[0], eta[1]
        p = eta[2] if len(eta) == 3 else 1
        a = (y2 - y1) / (x2 - x1) ** p
        etas = a * (np.clip(ts, x1, x2) - x1) ** p + y1
    else:
        etas = np.linspace(eta[0], eta[1], NUM_TRAIN)
    return np.clip(etas, 0, None)


def _env_on(name):
    """A/B switches: set and not "" / "0" """
    return os.environ.get(name, "0") not in ("", "0")


def attn_layer_selection(L, attn_res=None, attn_from_where=("up", "down")):
    """(res_div, layer mask) of etainv_maps_configure / etainv_maps_word_maps_ex for the reference's (attn_res, attn_from_where)"""
    attn_res = L // 4 if attn_res is None else int(attn_res)
    if attn_res <= 0 or L % attn_res or L // attn_res not in (2, 4, 8):
        raise NotImplementedError(f"attn_res must be L/2, L/4 or L/8 (cross layers exist at those sizes only), got {attn_res} at L = {L}")
    div = L // attn_res
    if div == 8:                                                # ptp.py:293-294 (`res == 8` at the reference's 64 x 64 latents)
        return div, 0x01
    where = set(attn_from_where)
    if not where <= {"up", "down", "mid"}:
        raise ValueError(f"attn_from_where entries must be 'up', 'down' or 'mid', got {sorted(where)}")
    mask = (0x03 if "down" in where else 0) | (0x1c if "up" in where else 0)
    if mask == 0:
        raise ValueError(f"attn_from_where {sorted(where)} has no cross layer with {attn_res}^2 tokens (the reference fails in torch.cat, ptp.py:301)")
    return div, mask


class EtaLoop:

    def __init__(self, engine, S=50, guidance_scale_bwd=7.5, guidance_scale_fwd=1.0, eta=(0.0, 0.4), noise_sample_count=10,
                 use_mask=True, mask_thres=0.2, skip_uncond_fwd=True, steps_offset=0, mask_eta="fwd_mean", mask_pow=None, target_dirinv=None,
                 mask_dirinv=None, skip_dead_source_rows=True, attn_res=None, attn_from_where=("up", "down")):
        self.e, self.S, self.L = engine, S, engine.L
        # which cross layers the eta mask's word maps average (mask_mode_cfg attn_res / attn_from_where, eta_inversion.py:161-162; aggregate_attention
        # ptp.py:288-303: the layers with res^2 tokens of the named locations, `res == 8` -> the mid block whatever from_where says)
        self.attn_div, self.attn_layer_mask = attn_layer_selection(self.L, attn_res, attn_from_where)
        if use_mask and self.attn_div != 4 and any(str(m).startswith("bwd") for m in (mask_eta, mask_dirinv)):
            raise NotImplementedError("bwd_* mask sources with attn_res != L/4: the backward pass keeps the (L/4)^2 store LocalBlend reads")
        # guidance_scale_fwd may be a (start, end) pair: linspace over the 1000 training timesteps, indexed by t (eta_inversion.py:108-110,325-326)
        self.g_fwd_table = np.linspace(guidance_scale_fwd[0], guidance_scale_fwd[1], NUM_TRAIN) if isinstance(guidance_scale_fwd, (tuple, list)) else None
        self.g_bwd, self.g_fwd = float(guidance_scale_bwd), (1.0 if self.g_fwd_table is not None else float(guidance_scale_fwd))
        self.ac = alphas_cumprod()
        self.delta = NUM_TRAIN // S
        self.t_bwd = ((np.arange(S) * self.delta)[::-1] + steps_offset).astype(np.int64)
        self.t_fwd = self.t_bwd[::-1].copy()
        self.etas = eta_table(eta)
        self.n_cand = noise_sample_count
        self.use_mask, self.mask_thres = use_mask, mask_thres
        # non-default eta-mask modes (reference eta_inversion.py:164-201): source of the map (fwd_mean | fwd | gt), thres None = soft, pow
        self.mask_eta, self.mask_pow = mask_eta, mask_pow
        # target_dirinv w: x_tgt += w (1 - mask_dirinv) (x_prev_src - x_src_new)  (eta_inversion.py:251-256); mask_dirinv names its own map
        # source (any of the mask_eta sources) or None
        self.target_dirinv, self.mask_dirinv = target_dirinv, mask_dirinv
        assert target_dirinv is None or use_mask, "target_dirinv is part of the masked update"
        # u + 1*(c - u) == c up to rounding: the uncond half of the forward pass is dead work when g_fwd == 1
        self.skip_uncond_fwd = skip_uncond_fwd and self.g_fwd == 1.0 and self.g_fwd_table is None
        # Backward steps with eta(t) == 0 (the whole ramp-free part of the paper's schedule: 30 of 50 steps with eta [[0.6, 0], [1, 0.7]]): the source
        # row is REPLAYED from the inversion trajectory (eta_inversion.py:247-249) and its guided noise feeds nothing but the best-of-n choice of a
        # noise sample that is then multiplied by eta sigma = 0 (:232, :330-375) -- eps(uncond source) is dead work.  Those steps run 3 B UNet rows
        # [u_t, c_s, c_t] with prompt-to-prompt (the cond source row stays: its attention probabilities are what is injected into the target) and
        # 2 B rows [u_t, c_t] without an attention coupling (simple editor); MasaCtrl couples u_t to u_s and keeps all four.  The source row then is
        # x_prev_src itself instead of x + (x_prev_src - x): at most one rounding apart (SURVEY E-11).  Like skip_uncond_fwd an exact identity of the
        # reference's arithmetic, not an approximation; skip_dead_source_rows=False runs the reference's row count.
        self.skip_dead_source_rows = skip_dead_source_rows and target_dirinv is None and not _env_on("ETAINV_NO_DEAD_ROW_SKIP")   # (env: A/B switch)
        # ... and once nothing is injected from the source (no cross replacement, self-replace over: steps >= 30 of 50 with the PIE settings) the cond source
        # row of such a step leaves the network after the last stored (L/4)^2 cross layer (transformer block 9; etainv_attn_ctrl.src_exit_block): its
        # noise prediction is unused.  ETAINV_NO_SRC_EXIT=1: A/B switch.
        self.src_exit = self.skip_dead_source_rows and not _env_on("ETAINV_NO_SRC_EXIT")
        self.src_exit_9_only = _env_on("ETAINV_SRC_EXIT_9_ONLY")   # A/B: no exit while the self-replace runs
        # share of a sample-forward's FLOPs a row has run when it leaves after transformer block 9 / 12 (layer walk at THIS latent size:
        # 0.509 / 0.716 at L = 64, 0.492 / 0.677 at L = 96 where the N^2 self-attention terms weigh more)
        self.SRC_EXIT_SHARE, self.SRC_EXIT_SHARE_12 = exit_share(self.L, 9), exit_share(self.L, 12)
        self.rows_executed = 0                                   # UNet sample-forwards issued by invert / sample since construction (bench accounting)
        self.lib = engine.lib

    def _alpha(self, tau):
        tau = min(int(tau), NUM_TRAIN - 1)
        return float(self.ac[tau]) if tau >= 0 else float(self.ac[0])

    # ---------------------------------------------------------------- forward / inversion
    def invert(self, z0, ctx_src, tokens=None, teacher=None):
        """z0 (B,4,L,L) fp32; ctx_src (B,2,77,768) = [uncond, cond] per image; tokens (B,W) int32 token index of each
        whitespace word (first occurrence + 1).  Returns latents (S+1,B,4,L,L) and the mean word maps (B,W,L,L).
        teacher (S+1,B,4,L,L), tests only: step j reads teacher[j] instead of its own previous output (teacher-forced parity:
        every step is compared with the oracle on the oracle's input, so rounding is not amplified by the recursion)."""
        e, S, L = self.e, self.S, self.L
        B = z0.shape[0]
        dev = z0.device
        lat = torch.empty(S + 1, B, 4, L, L, dtype=torch.float32, device=dev)
        lat[0].copy_(z0)
        if self.skip_uncond_fwd:
            ctx = ctx_src[:, 1].contiguous().float()
        else:
            ctx = torch.cat([ctx_src[:, 0], ctx_src[:, 1]]).contiguous().float()
        rows = ctx.shape[0]
        eps_all = torch.empty(rows, 4, L, L, dtype=torch.float32, device=dev)
        eps = eps_all if self.skip_uncond_fwd else torch.empty(B, 4, L, L, dtype=torch.float32, device=dev)
        maps_mean = maps_steps = None
        ctrl = None
        if self.use_mask:
            assert tokens is not None
            maps_mean = torch.zeros(B, tokens.shape[1], L, L, dtype=torch.float32, device=dev)
            if "fwd" in (self.mask_eta, self.mask_dirinv):                        # per-step maps, keyed by step (eta_inversion.py:44-49,168)
                maps_steps = torch.zeros(S, B, tokens.shape[1], L, L, dtype=torch.float32, device=dev)
            ctrl = AttnControl(mode=_capi.ATTN_STORE, n_img=B, store_maps=True)
            if e.map_div != self.attn_div:
                e.maps_configure(self.attn_div)                                   # (clears the store)
            else:
                e.maps_reset()
        n = B * 4 * L * L
        st = _capi.stream_ptr()
        with e.cached_context():                               # one unchanged context tensor for all S calls
            for j, t in enumerate(self.t_fwd):
                x_in = lat[j] if teacher is None else teacher[j].contiguous()
                e.unet(x_in, int(t), ctx, ctrl, out=eps_all)
                self.rows_executed += rows
                if not self.skip_uncond_fwd:
                    g = float(self.g_fwd_table[int(t)]) if self.g_fwd_table is not None else self.g_fwd
                    _capi.check(self.lib.etainv_cfg_combine(_capi.ptr(eps_all[:B]), _capi.ptr(eps_all[B:]), g, _capi.ptr(eps), n,
                                                            _capi.F32, st))
                a_from, a_to = self._alpha(int(t) - self.delta), self._alpha(int(t))   # "sameshift" (scheduling_ddim_inverse.py:127-131)
                _capi.check(self.lib.etainv_ddim_step(_capi.ptr(x_in), _capi.ptr(eps), a_from, a_to, _capi.ptr(lat[j + 1]), n, _capi.F32, st))
                if self.use_mask:
                    e.word_maps_ex(B, tokens, j + 1, 0, self.attn_layer_mask, maps_mean, accumulate=True, scale=1.0 / S)
                    if maps_steps is not None:
                        e.word_maps_ex(B, tokens, j + 1, 0, self.attn_layer_mask, maps_steps[j], accumulate=False, scale=1.0)
        return {"latents": lat, "maps_mean": maps_mean, "maps_steps": maps_steps}

    # ---------------------------------------------------------------- backward / eta sampling
    def sample(self, inv, ctx_src, ctx_tgt, noise, edit_word=None, ptp=None, masactrl=None, trace=None, gt_mask=None, edit_word_tgt=None,
               teacher=None):
        """noise (S,n_cand,4,L,L) fp32: the candidates of every step (reference draws them from a generator reseeded
        per image, eta_inversion.py:156,276, so all images share the table).  edit_word (B,) index into the word maps.
        ptp: PtpTables or None; masactrl: (start_step, first_block) or None.  Returns latents (2B,4,L,L) [src.., tgt..].
        teacher (S,2B,4,L,L), tests only: step i starts from teacher[i] (see invert)."""
        e, S, L = self.e, self.S, self.L
        lat_inv = inv["latents"]
        B = lat_inv.shape[1]
        dev = lat_inv.device
        ctx = torch.cat([ctx_src[:, 0], ctx_tgt[:, 0], ctx_src[:, 1], ctx_tgt[:, 1]]).contiguous().float()   # [u_s,u_t,c_s,c_t] x B
        x = torch.cat([lat_inv[S], lat_inv[S]]).contiguous()
        x_new = torch.empty_like(x)
        eps_all = torch.empty(4 * B, 4, L, L, dtype=torch.float32, device=dev)
        best = torch.zeros(B, dtype=torch.int32, device=dev)
        scratch = torch.empty(B * 16 * 64, dtype=torch.float32, device=dev)
        mask_map, mask_mode = None, int(self.use_mask)
        src_map = shaped = None
        if self.use_mask:
            idx = edit_word.to(dev).long().reshape(B, 1, 1, 1).expand(B, 1, L, L)
            final = self.mask_thres is None or self.mask_pow is not None or self.mask_eta != "fwd_mean"
            mask_mode = 2 if final else 1                                           # 2: the map is the per-pixel eta multiplier
            sources = {self.mask_eta, self.mask_dirinv} - {None}
            if any(sname.startswith("bwd") for sname in sources):                    # maps of the backward-pass store (eta_inversion.py:176-183)
                assert ptp is not None, "bwd_* masks read the prompt-to-prompt controller's maps"
                tok_s = (edit_word.to(dev).int() + 1).reshape(B, 1).contiguous()
                tok_t = ((edit_word_tgt if edit_word_tgt is not None else edit_word).to(dev).int() + 1).reshape(B, 1).contiguous()
                map_s = torch.empty(B, 1, L, L, dtype=torch.float32, device=dev)
                map_t = torch.empty(B, 1, L, L, dtype=torch.float32, device=dev)
            static = {}
            if "gt" in sources:
                assert gt_mask is not None, "a 'gt' mask source needs the ground-truth mask (B,L,L)"
                static["gt"] = gt_mask.to(dev).float().reshape(B, L, L)
            if "fwd_mean" in sources:
                static["fwd_mean"] = inv["maps_mean"].gather(1, idx).reshape(B, L, L)

            def shaped(m):                                                          # get_mask tail, eta_inversion.py:196-201
                if self.mask_thres is not None:
                    m = (m > self.mask_thres).to(m.dtype)
                if self.mask_pow is not None:
                    m = torch.pow(m, self.mask_pow)
                return m.contiguous()

            def src_map(name, i):                                                   # raw map of one source at backward step i (eta_inversion.py:164-183)
                if name in static:
                    return static[name]
                if name == "fwd":                                                   # map of THIS timestep (t_bwd[i] == t_fwd[S-1-i])
                    return inv["maps_steps"][S - 1 - i].gather(1, idx).reshape(B, L, L)
                if name != "bwd_target":                                            # average over the i+1 backward steps done, this one included
                    e.word_maps_ex(B, tok_s, i + 1, 0, self.attn_layer_mask, map_s)
                if name != "bwd_source":
                    e.word_maps_ex(B, tok_t, i + 1, 1, self.attn_layer_mask, map_t)
                m = map_s if name == "bwd_source" else map_t if name == "bwd_target" else torch.maximum(map_s, map_t)
                return m.reshape(B, L, L)
        if ptp is not None:
            if e.map_div != 4:
                e.maps_configure(4)                                                  # LocalBlend and the bwd_* sources read the (L/4)^2 layers
            else:
                e.maps_reset()
        ctx3 = eps3 = ctx3x = eps3x = None
        losses = torch.zeros(B, self.n_cand, dtype=torch.float32, device=dev) if trace is not None else None   # per-candidate losses of the best-of-n step (traces only)
        eps_t = torch.empty(B, 4, L, L, dtype=torch.float32, device=dev)
        st = _capi.stream_ptr()
        with e.cached_context():                               # one unchanged context tensor for all S calls
            for i, t in enumerate(self.t_bwd):
                t = int(t)
                ctrl = None
                if ptp is not None:
                    live = bool(ptp.cross_active[i])
                    ctrl = AttnControl(mode=_capi.ATTN_PTP, n_img=B, store_maps=True, mapper=ptp.mapper if live else None, alphas=ptp.alphas,
                                       replace_mat=ptp.replace_mat if live else None, equalizer=ptp.equalizer, cross_alpha=ptp.cross_alpha[i],
                                       self_replace_active=ptp.self_lo <= i < ptp.self_hi, self_max_tokens=(L // 2) ** 2)
                elif masactrl is not None:
                    ctrl = AttnControl(mode=_capi.ATTN_MASA, n_img=B, masa_active=masactrl[0] <= i < 50, masa_first_block=masactrl[1])
                if teacher is not None:
                    x.copy_(teacher[i])
                p = t - self.delta
                a_t, a_p = float(self.ac[t]), (float(self.ac[p]) if p >= 0 else float(self.ac[0]))
                var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
                if self.skip_dead_source_rows and float(self.etas[t]) == 0.0 and masactrl is None:
                    # eta == 0: no eps(uncond source) -- rows [u_t, c_s, c_t] over latents [tgt, src] (ptp) or [u_t, c_t] over [tgt] (no coupling)
                    if ptp is not None and not live and self.src_exit and not (self.src_exit_9_only and ptp.self_lo <= i < ptp.self_hi):
                        # no cross replacement any more: the cond source row only feeds the AttentionStore of the five (L/4)^2 cross layers (LocalBlend /
                        # bwd_* masks; last one = block 9) and, while the self-replace runs, the (L/2)^2-token self-attentions (last one = block 12) --
                        # rows [u_t, c_t, c_s], c_s leaves after that block
                        if ctx3x is None:
                            ctx3x = torch.cat([ctx_tgt[:, 0], ctx_tgt[:, 1], ctx_src[:, 1]]).contiguous().float()
                            eps3x = torch.empty(3 * B, 4, L, L, dtype=torch.float32, device=dev)
                        self_on = ptp.self_lo <= i < ptp.self_hi
                        ctrl.c.first_row, ctrl.c.src_exit_block = B, 12 if self_on else 9
                        e.unet(torch.cat([x[B:], x[B:], x[:B]]), t, ctx3x, ctrl, out=eps3x)
                        eu, ec = eps3x[:B], eps3x[B:2 * B]
                        rows_out, layout = eps3x[:2 * B], "u_t,c_t"                    # (the exited c_s rows have no output)
                        self.rows_executed += 2 * B + B * (self.SRC_EXIT_SHARE_12 if self_on else self.SRC_EXIT_SHARE)   # (share of the UNet's FLOPs the exited rows ran)
                    elif ptp is not None:
                        if ctx3 is None:
                            ctx3 = torch.cat([ctx_tgt[:, 0], ctx_src[:, 1], ctx_tgt[:, 1]]).contiguous().float()
                            eps3 = torch.empty(3 * B, 4, L, L, dtype=torch.float32, device=dev)
                        ctrl.c.first_row = B
                        e.unet(torch.cat([x[B:], x[:B]]), t, ctx3, ctrl, out=eps3)
                        eu, ec = eps3[:B], eps3[2 * B:]
                        rows_out, layout = eps3, "u_t,c_s,c_t"
                        self.rows_executed += 3 * B
                    else:
                        if ctx3 is None:
                            ctx3 = torch.cat([ctx_tgt[:, 0], ctx_tgt[:, 1]]).contiguous().float()
                            eps3 = torch.empty(2 * B, 4, L, L, dtype=torch.float32, device=dev)
                        e.unet(x[B:], t, ctx3, None, out=eps3)
                        eu, ec = eps3[:B], eps3[B:]
                        rows_out, layout = eps3, "u_t,c_t"
                        self.rows_executed += 2 * B
                    n_t = B * 4 * L * L
                    _capi.check(self.lib.etainv_cfg_combine(_capi.ptr(eu), _capi.ptr(ec), self.g_bwd, _capi.ptr(eps_t), n_t, _capi.F32, st))
                    _capi.check(self.lib.etainv_ddim_eta_step(_capi.ptr(x[B:]), _capi.ptr(eps_t), 0.0, None, 0, None, a_t, a_p, var, B, 4, L * L,
                                                              _capi.ptr(x_new[B:]), _capi.F32, st))
                    x_new[:B].copy_(lat_inv[S - 1 - i])                                 # the replayed source row
                    best.zero_()                                                        # (the reference's argmin over NaN losses: index 0)
                    x, x_new = x_new, x
                    if ptp is not None and ptp.blend_alpha is not None and (i + 1) > int(0.2 * S):
                        e.local_blend(x, B, ptp.blend_alpha, 0.3)
                    if trace is not None:
                        # the rows this step executed, with their layout (B rows per name); "eps_all" (the 4 B-row layout) does not exist here
                        trace.append({"t": t, "latent": x.clone(), "best": best.clone(), "eps_all": None, "eps_rows": rows_out.clone(), "layout": layout})
                    continue
                e.unet(x, t, ctx, ctrl, out=eps_all)
                self.rows_executed += 4 * B
                dmap = None
                if self.use_mask:
                    raw = src_map(self.mask_eta, i)
                    mask_map = shaped(raw) if mask_mode == 2 else raw.contiguous()      # mode 1: the kernel thresholds the raw forward-mean map
                    if self.target_dirinv is not None and self.mask_dirinv is not None:  # 1 - shaped map of the mask_dirinv source (eta_inversion.py:234-256)
                        dmap = (1.0 - shaped(raw if self.mask_dirinv == self.mask_eta else src_map(self.mask_dirinv, i))).contiguous()
                _capi.check(self.lib.etainv_eta_backward_step_ex(
                    _capi.ptr(x), _capi.ptr(eps_all), self.g_bwd, _capi.ptr(lat_inv[S - 1 - i]), _capi.ptr(noise[i]), self.n_cand,
                    float(self.etas[t]), _capi.ptr(mask_map), float(self.mask_thres or 0.0), mask_mode, a_t, a_p, var, B, 4, L * L,
                    _capi.ptr(x_new), None, _capi.ptr(best), _capi.ptr(losses), _capi.ptr(scratch), _capi.F32, float(self.target_dirinv or 0.0), _capi.ptr(dmap), st))
                x, x_new = x_new, x
                if ptp is not None and ptp.blend_alpha is not None and (i + 1) > int(0.2 * S):
                    e.local_blend(x, B, ptp.blend_alpha, 0.3)                       # LocalBlend, reference ptp.py:31-47
                if trace is not None:
                    trace.append({"t": t, "latent": x.clone(), "best": best.clone(), "eps_all": eps_all.clone(), "eps_rows": eps_all.clone(),
                                  "layout": "u_s,u_t,c_s,c_t", "losses": losses.clone()})
        return x


class PtpTables:
    """Device tables of one prompt-to-prompt edit per image (what ptp.make_controller builds on the host,
    reference modules/utils/ptp.py:306-320)."""

    def __init__(self, mapper, alphas, cross_alpha, self_replace_steps, S, equalizer=None, blend_alpha=None, replace_mat=None,
                 device="cuda"):
        T = lambda a, dt: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device).contiguous()
        self.mapper = T(mapper, torch.int32)              # (B,77)
        self.alphas = T(alphas, torch.float32)            # (B,77)
        self.equalizer = T(equalizer, torch.float32)      # (B,77)
        self.replace_mat = T(replace_mat, torch.float32)  # (B,77,77)
        self.blend_alpha = T(blend_alpha, torch.float32)  # (B,2,77)
        self.cross_alpha = T(cross_alpha, torch.float32)  # (S+1,B,77)
        # steps whose cross_replace_alpha row is all zero: the cross edit is the identity there (rep * 0 + 1 * own, reference ptp.py:228) -- the loop
        # then runs the plain cross-attention launch for the cond-target rows too (no source-probability recomputation); the map store stays on
        ca = np.asarray(cross_alpha, dtype=np.float32)
        self.cross_active = (ca.reshape(ca.shape[0], -1) != 0).any(1)
        if isinstance(self_replace_steps, float):
            self_replace_steps = (0, self_replace_steps)
        self.self_lo, self.self_hi = int(S * self_replace_steps[0]), int(S * self_replace_steps[1])


def noise_table(S, n, L, seed=0, device="cuda"):
    """CPU torch generator, reseeded per image like the reference (eta_inversion.py:276): one table for all images."""
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randn((n, 1, 4, L, L), generator=g) for _ in range(S)]).reshape(S, n, 4, L, L).to(device)
