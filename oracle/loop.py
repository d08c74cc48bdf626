"""Forward (inversion) and backward (eta-sampling) loops of `etainv` with the simple / ptp /
masactrl editors, restated for the CPU oracle (test infrastructure).

Follows (reference file:line):
  * predict_noise (always uncond+cond, CFG)        modules/inversion/eta_inversion.py:319-328
  * forward loop                                   modules/inversion/diffusion_inversion.py:314-341, 388-418
  * invert + per-step word maps + mean             modules/inversion/eta_inversion.py:36-49, 378-404
  * best-of-n variance noise                       modules/inversion/eta_inversion.py:296-317, 330-375
  * masked-eta backward step + source replay       modules/inversion/eta_inversion.py:159-273
  * backward loop                                  modules/inversion/eta_inversion.py:275-294
  * batch layout [u_s,u_t,c_s,c_t], latent x2      modules/inversion/diffusion_inversion.py:462-528
  * editors                                        modules/editing/editor.py:67-118, simple_editor.py:27-51,
                                                   masactrl_editor.py:44-69
  * attention hook (materialised probabilities)    modules/utils/ptp_utils.py:205-260
  * MasaCtrl mutual self-attention                 modules/utils/masactrl.py:41-72, masactrl_utils.py:18-31
"""
import numpy as np
import torch

from . import schedule as sch
from . import ptp as optp


# --------------------------------------------------------------------------- attention hooks
def _plain(q, k, v, scale):
    return (torch.einsum("bid,bjd->bij", q, k) * scale).softmax(dim=-1)


def _merge_heads(out, heads):
    bh, n, d = out.shape
    return out.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, d * heads)


def ptp_hook(controller):
    """ptp_utils.py:238-258: probs -> controller -> probs @ v.
    Self-attention maps with more than `store_max_n` (32^2) query tokens are neither stored nor edited by
    any controller (ptp.py:153-157,195-199), so the oracle does not materialise them: it only advances
    the controller's layer counter -- same result, minutes faster on CPU."""
    def ctrl(is_cross, layer_idx, place, q, k, v, scale, heads):
        if not is_cross and q.shape[1] > controller.store_max_n and q.shape[1] > getattr(controller, "thres_n", 0):
            out = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=scale)
            controller.count_layer()
            return _merge_heads(out, heads)
        attn = _plain(q, k, v, scale)
        attn = controller(attn, is_cross, place)
        return _merge_heads(torch.einsum("bij,bjd->bid", attn, v), heads)
    return ctrl


class MasaCtrl:
    """MutualSelfAttentionControl(start_step=4, start_layer=10, total_steps=50): masactrl.py:20-72."""

    def __init__(self, start_step=4, start_layer=10, total_steps=50, total_layers=16, num_att_layers=32, fast_n=2304):
        self.step_idx = list(range(start_step, total_steps))
        self.layer_idx = list(range(start_layer, total_layers))
        self.cur_step, self.cur_att_layer, self.num_att_layers = 0, 0, num_att_layers
        # more than fast_n query tokens: the same softmax(q k^T) v through torch's fused SDPA, so that the 9216 x 9216 probabilities
        # of a 768^2 image (5.4 GB per half in fp32) are never materialised on the CPU; checked against the literal path in
        # tests/test_oracle_golden.py
        self.fast_n = fast_n

    def __call__(self, is_cross, layer_idx, place, q, k, v, scale, heads):
        active = (not is_cross) and self.cur_step in self.step_idx and (self.cur_att_layer // 2) in self.layer_idx
        sdpa = torch.nn.functional.scaled_dot_product_attention
        if not active:
            if q.shape[1] > self.fast_n:
                out = _merge_heads(sdpa(q, k, v, scale=scale), heads)
            else:
                out = _merge_heads(torch.einsum("bij,bjd->bid", _plain(q, k, v, scale), v), heads)
        elif q.shape[1] > self.fast_n:
            outs = []
            for qh, kh, vh in zip(q.chunk(2), k.chunk(2), v.chunk(2)):      # uncond half, cond half: K, V of the half's source sample
                b = qh.shape[0] // heads
                o = sdpa(qh.reshape(b, heads, *qh.shape[1:]), kh[:heads][None], vh[:heads][None], scale=scale)
                outs.append(o.permute(0, 2, 1, 3).reshape(b, qh.shape[1], -1))
            out = torch.cat(outs, dim=0)
        else:
            outs = []
            for qh, kh, vh in zip(q.chunk(2), k.chunk(2), v.chunk(2)):      # uncond half, cond half
                ks, vs = kh[:heads], vh[:heads]                             # source sample's K, V
                b = qh.shape[0] // heads
                qq = qh.reshape(b, heads, *qh.shape[1:])
                a = (torch.einsum("bhid,hjd->bhij", qq, ks) * scale).softmax(-1)
                o = torch.einsum("bhij,hjd->bhid", a, vs)
                outs.append(o.permute(0, 2, 1, 3).reshape(b, qh.shape[1], -1))
            out = torch.cat(outs, dim=0)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
        return out


# --------------------------------------------------------------------------- the loops
class EtaInversionOracle:
    def __init__(self, unet, S=50, guidance_scale_bwd=7.5, guidance_scale_fwd=1, eta=(0.0, 0.4),
                 noise_sample_count=10, use_mask=True, thres=0.2, L=64, dtype=torch.float32, mask_eta="fwd_mean", mask_pow=None, target_dirinv=None, mask_dirinv=None,
                 attn_res=None, attn_from_where=("up", "down")):
        self.unet, self.S, self.L, self.dtype = unet, S, L, dtype
        self.g_bwd, self.g_fwd = guidance_scale_bwd, guidance_scale_fwd
        if isinstance(guidance_scale_fwd, (tuple, list)):                     # per-timestep table, eta_inversion.py:108-110,325-326
            self.g_fwd = np.linspace(guidance_scale_fwd[0], guidance_scale_fwd[1], 1000)
        self.ac = sch.alphas_cumprod()
        self.t_fwd, self.t_bwd = sch.timesteps_forward(S), sch.timesteps_backward(S)
        self.etas = sch.eta_table(eta)
        self.n = noise_sample_count
        self.use_mask, self.thres = use_mask, thres
        self.mask_eta, self.mask_pow = mask_eta, mask_pow     # eta_inversion.py:164-201 (gt / fwd / fwd_mean; thres None; pow)
        self.target_dirinv, self.mask_dirinv = target_dirinv, mask_dirinv   # eta_inversion.py:251-256
        self.attn_res = attn_res if attn_res is not None else L // 4     # 16 at L=64 (eta_inversion.py:90); mask_mode_cfg["attn_res"] (:161)
        self.attn_from_where = tuple(attn_from_where)                      # mask_mode_cfg["attn_from_where"] (:162)
        self.thres_n = (L // 2) ** 2            # 32^2 at L=64 (ptp.py:153,226)

    # eta_inversion.py:319-328
    def predict_noise(self, latent, t, context, g):
        x = torch.cat([latent] * 2) if latent.shape[0] != context.shape[0] else latent
        out = self.unet(x, torch.tensor(int(t)), encoder_hidden_states=context)["sample"]
        u, c = out.chunk(2)
        return u + g * (c - u)

    # eta_inversion.py:378-404 + diffusion_inversion.py:388-418
    def invert(self, z0, context, prompt, teacher=None):
        """context (2,77,768) = [uncond, cond]; one map per whitespace word of `prompt`; a repeated word
        reuses the token of its FIRST occurrence (`prompt.split(' ').index(word)`, ptp_editor.py:72).
        teacher (list of S+1 latents, precision-floor runs only): step j starts from teacher[j] instead of this run's own latent."""
        words = prompt.split(" ")
        n_words = len(words)
        tok_idx = [words.index(w) + 1 for w in words]
        store = None
        if self.use_mask:
            store = optp.AttentionStore(store_max_n=self.thres_n)
            self.unet.set_ctrl(ptp_hook(store))
        latent = z0.clone()
        latents, noise_preds, maps_per_t = [z0], [], {}
        try:
            for j, t in enumerate(self.t_fwd):
                if teacher is not None:
                    latent = teacher[j].clone()
                eps = self.predict_noise(latent, t, context, float(self.g_fwd[int(t)]) if isinstance(self.g_fwd, np.ndarray) else self.g_fwd)
                a_from, a_to = sch.ddim_inverse_coeffs(self.ac, int(t), self.S)
                latent = sch.ddim_step(latent, eps, a_from, a_to)
                if store is not None:                                     # eta_inversion.py:44-49
                    maps_per_t[int(t)] = [
                        optp.attention_map(store, ti, res=self.attn_res, from_where=self.attn_from_where, resize=self.L)
                        for ti in tok_idx]
                noise_preds.append(eps)
                latents.append(latent)
        finally:
            self.unet.set_ctrl(None)
        res = {"latents": latents, "noise_preds": noise_preds, "zT_inv": latents[-1], "context": context}
        if store is not None:                                             # eta_inversion.py:392-396
            lst = list(maps_per_t.values())
            res["attn_maps_mean"] = [torch.stack([a[w] for a in lst]).mean(0) for w in range(n_words)]
            res["attn_maps_per_t"] = maps_per_t
        return res

    # eta_inversion.py:330-375
    def eta_variance_noise(self, latent_prev, latent, t, noise_pred, noise_choices):
        eta = float(self.etas[int(t)])
        mean = sch.ddim_eta_step(latent, noise_pred, self.ac, int(t), self.S, eta, noise=None)
        std = eta * sch.variance(self.ac, int(t), self.S) ** 0.5
        with np.errstate(all="ignore"):
            opt = (latent_prev - mean) / torch.tensor(std, dtype=latent.dtype)
        losses = torch.square(noise_choices - opt).reshape(noise_choices.shape[0], -1).mean(1)
        best = int(torch.argmin(losses).item())
        return eta, noise_choices[best], best, losses

    # eta_inversion.py:159-205: raw map of one mask source (before thres / pow)
    def source_map(self, source, t, inv, gt_mask, controller):
        ew = self._edit_word_idx
        if source == "gt":
            return gt_mask                                                    # already at latent resolution (eta_inversion.py:286-287)
        if source == "fwd":
            return inv["attn_maps_per_t"][int(t)][ew[0]]                      # eta_inversion.py:168
        if source == "fwd_mean":
            return inv["attn_maps_mean"][ew[0]]                               # eta_inversion.py:171
        # eta_inversion.py:176-183: maps of the backward-pass controller, averaged over the steps done so far (this one included)
        amap = lambda word, sel: optp.attention_map(controller, word + 1, res=self.attn_res, from_where=self.attn_from_where, resize=self.L,
                                                    num_prompts=2, select=sel)
        if source == "bwd_source":
            return amap(ew[0], 0)
        if source == "bwd_target":
            return amap(ew[1], 1)
        assert source == "bwd_source_target", source
        return torch.maximum(amap(ew[0], 0), amap(ew[1], 1))

    def _shape_mask(self, m):
        if self.thres is not None:
            m = (m > self.thres).to(m.dtype)                                  # eta_inversion.py:196-198
        if self.mask_pow is not None:
            m = torch.pow(m, self.mask_pow)                                   # eta_inversion.py:200-201
        return m

    # eta_inversion.py:207-273
    def step_backward(self, latent, t, context, source_latent_prev, noise_choices, mask_map, controller, dirinv_map=None, inv=None, gt_mask=None):
        """mask_map / dirinv_map: raw maps of the mask_eta / mask_dirinv sources for sources that do not depend on this step's UNet call
        (gt, fwd, fwd_mean); bwd_* sources are read from the controller AFTER the UNet call, like the reference (get_mask runs after
        predict_noise, eta_inversion.py:225-237).  dirinv_map None with a mask_dirinv configured = the same map as mask_eta."""
        eps = self.predict_noise(latent, t, context, self.g_bwd)
        if self.use_mask and self.mask_eta.startswith("bwd"):
            mask_map = self.source_map(self.mask_eta, t, inv, gt_mask, controller)
        if self.use_mask and self.mask_dirinv is not None and self.mask_dirinv.startswith("bwd"):
            dirinv_map = self.source_map(self.mask_dirinv, t, inv, gt_mask, controller)
        eta, z, best, losses = self.eta_variance_noise(source_latent_prev, latent[:1], t, eps[:1], noise_choices)
        eta_map = torch.full_like(z, eta)
        if self.use_mask:
            m = self._shape_mask(mask_map)
            eta_map = m * eta_map
            new = sch.ddim_eta_step(latent, eps, self.ac, int(t), self.S, eta_map, noise=z)
            delta = source_latent_prev[:1] - new[:1]
            new[:1] = new[:1] + delta                                      # eta_inversion.py:247-249
            if self.target_dirinv is not None:                             # eta_inversion.py:251-256
                if self.mask_dirinv is not None:
                    md = m if dirinv_map is None else self._shape_mask(dirinv_map)
                    delta = (1 - md) * delta
                new[1:] = new[1:] + self.target_dirinv * delta
        else:
            new = sch.ddim_eta_step(latent, eps, self.ac, int(t), self.S, eta_map, noise=z)
            new[:1] = source_latent_prev[:1]
        new = new.clone()
        if controller is not None:
            new = controller.step_callback(new)                           # ptp_editor.py:92-98
        return new, eps, best, losses

    # diffusion_inversion.py:493-528 + eta_inversion.py:275-294
    def sample(self, inv, ctx_src, ctx_tgt, noise_table, edit_word_idx=None, controller=None, masactrl=None,
               trace=None, gt_mask=None, teacher=None):
        """noise_table: (S, n, 1, 4, L, L) -- the candidates `sample_variance_noise` would draw at each
        step from the per-image generator (eta_inversion.py:156,276), injected for reproducibility.
        teacher (list of S latents (2,4,L,L), precision-floor runs only): step i starts from teacher[i]."""
        context = torch.stack([ctx_src, ctx_tgt], 1).reshape(4, *ctx_src.shape[1:])   # [u_s,u_t,c_s,c_t]
        self._edit_word_idx = edit_word_idx
        latent = torch.cat([inv["latents"][-1]] * 2)
        if controller is not None:
            self.unet.set_ctrl(ptp_hook(controller))
        elif masactrl is not None:
            self.unet.set_ctrl(masactrl)
        try:
            for i, t in enumerate(self.t_bwd):
                if teacher is not None:
                    latent = teacher[i].clone()
                mask_map = dirinv_map = None
                if self.use_mask and not self.mask_eta.startswith("bwd"):
                    mask_map = self.source_map(self.mask_eta, t, inv, gt_mask, controller)
                if self.use_mask and self.mask_dirinv is not None and self.mask_dirinv != self.mask_eta and not self.mask_dirinv.startswith("bwd"):
                    dirinv_map = self.source_map(self.mask_dirinv, t, inv, gt_mask, controller)
                latent, eps, best, losses = self.step_backward(
                    latent, t, context, inv["latents"][-(i + 2)], noise_table[i].to(latent.dtype), mask_map, controller,
                    dirinv_map=dirinv_map, inv=inv, gt_mask=gt_mask)
                if trace is not None:
                    trace.append({"t": int(t), "latent": latent.clone(), "eps": eps.clone(), "best": best,
                                  "losses": losses.clone()})
        finally:
            self.unet.set_ctrl(None)
        return latent


class DiffusionInversionOracle:
    """`diffinv`: plain DDIM inversion and deterministic DDIM sampling, no source replay (reference
    modules/inversion/diffusion_inversion.py:314-341 predict_step_forward, :343-371 predict_step_backward, :388-436 loops, :493-528
    sample).  `step_fwd` / `step_bwd` default to the DDIM closed forms; the DPM-Solver++ inverse pair plugs in through them."""

    def __init__(self, unet, S=50, guidance_scale_bwd=7.5, guidance_scale_fwd=1, step_fwd=None, step_bwd=None):
        self.unet, self.S, self.g_bwd, self.g_fwd = unet, S, guidance_scale_bwd, guidance_scale_fwd
        self.ac = sch.alphas_cumprod()
        self.t_fwd, self.t_bwd = sch.timesteps_forward(S), sch.timesteps_backward(S)
        self.step_fwd, self.step_bwd = step_fwd, step_bwd

    def predict_noise(self, latent, t, context, g):
        x = torch.cat([latent] * 2) if latent.shape[0] != context.shape[0] else latent
        u, c = self.unet(x, torch.tensor(int(t)), encoder_hidden_states=context)["sample"].chunk(2)
        return u + g * (c - u)

    def invert(self, z0, context):
        latent, latents = z0.clone(), [z0]
        for i, t in enumerate(self.t_fwd):
            eps = self.predict_noise(latent, t, context, self.g_fwd)
            if self.step_fwd is not None:
                latent = self.step_fwd(eps, int(t), latent, i)
            else:
                latent = sch.ddim_step(latent, eps, *sch.ddim_inverse_coeffs(self.ac, int(t), self.S))
            latents.append(latent)
        return {"latents": latents, "zT_inv": latents[-1]}

    def sample(self, inv, contexts):
        """contexts: list of (2,77,768) [uncond, cond] -- [source, target] or [target] (no_source_backward, editor.py:108-116)"""
        n = len(contexts)
        context = torch.stack(contexts, 1).reshape(2 * n, *contexts[0].shape[1:])     # cat_context (diffusion_inversion.py:438-460)
        latent = torch.cat([inv["latents"][-1]] * n)
        for i, t in enumerate(self.t_bwd):
            eps = self.predict_noise(latent, t, context, self.g_bwd)
            if self.step_bwd is not None:
                latent = self.step_bwd(eps, int(t), latent, i)
            else:
                latent = sch.ddim_eta_step(latent, eps, self.ac, int(t), self.S, 0.0, noise=None)
        return latent


def noise_table(S, n, L, seed=0):
    """eta_inversion.py:156,276 on CPU: one generator per image, n candidates drawn per step."""
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randn((n, 1, 4, L, L), generator=g) for _ in range(S)])
